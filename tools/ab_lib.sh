#!/bin/bash
# A/B of library builds on the GPU box: bench.py per config with FSGS_LIB pointing at each build.
# usage: bash tools/ab_lib.sh TAG "2 3" "libfsgs.so libfsgs_a.so ..."
TAG=${1:-ablib}; CFGS=${2:-"2"}; LIBS=${3:-"libfsgs.so"}; O=gpurun_out/$TAG; mkdir -p $O
for cfg in $CFGS; do for l in $LIBS; do
  FSGS_LIB=$PWD/fusionsense_amd/$l python bench.py --config $cfg --no-cpu-baseline --no-dropin > $O/c${cfg}_$l.json 2> $O/c${cfg}_$l.err
  python - "$O/c${cfg}_$l.json" "c$cfg $l" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = {n: v["avg_ms"] for n, v in d["kernels_ms"].items()}
    print(f"{sys.argv[2]:24s} {d['value']:9.2f} it/s {d['ms_per_step']:.4f} ms  p50 {d['gpu_step_ms']['p50']:.4f} | " + " ".join(f"{n.replace('raster_','r_').replace('_quad_d4e3','')}={v:.4f}" for n, v in sorted(k.items())))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done; done 2>&1 | tee $O/summary.txt
