#!/bin/bash
# A/B of one environment switch on the GPU box: bench.py per config with VAR set to each value.
# usage: bash tools/ab_env.sh TAG "4 5" FSGS_OCCLUSION_CUT "0 1" [extra bench args]
TAG=${1:-abenv}; CFGS=${2:-"2"}; VAR=$3; VALS=$4; shift 4; O=gpurun_out/$TAG; mkdir -p $O
for cfg in $CFGS; do for v in $VALS; do
  env $VAR=$v python bench.py --config $cfg --no-cpu-baseline --no-dropin "$@" > $O/c${cfg}_$v.json 2> $O/c${cfg}_$v.err
  python - "$O/c${cfg}_$v.json" "c$cfg $VAR=$v" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = {n: v["avg_ms"] for n, v in d["kernels_ms"].items()}
    oc = d.get("occlusion_cut", {})
    print(f"{sys.argv[2]:28s} {d['value']:9.2f} it/s {d['ms_per_step']:.4f} ms  p50 {d['gpu_step_ms']['p50']:.4f} cut {oc.get('cut_frames')}/{oc.get('cut_redone')} ovf {d.get('live_list_overflows')} | " + " ".join(f"{n.replace('raster_','r_').replace('_quad_d4e3','')}={v:.4f}" for n, v in sorted(k.items())))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done; done 2>&1 | tee $O/summary.txt
