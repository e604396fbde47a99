#!/bin/bash
# A/B of the compositing backward on the GPU box (through gpurun): bench.py under the switch of csrc/raster_live.hip
# (FSGS_BWD_MERGE_THR16: a segment walks the union list with merged atomics when its longest row list is >= thr/16 of
# the union; 0 = always, 17 = never).   usage: bash tools/ab_bwd.sh TAG "2 3" "17 15 14 12 8 0"
TAG=${1:-ab}; CFGS=${2:-"2 3"}; THRS=${3:-"17 15 14 12 8 0"}; O=gpurun_out/$TAG; mkdir -p $O
run() {  # name, config, env...
  local name=$1 cfg=$2; shift 2
  env "$@" python bench.py --config $cfg --no-cpu-baseline --no-dropin > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernels_ms"]
    b = [v for n, v in k.items() if n.startswith("raster_bwd")]
    f = [v for n, v in k.items() if n.startswith("raster_fwd")]
    g = k.get("gaussian_bwd", {}).get("avg_ms")
    print(f"{sys.argv[2]:28s} {d['value']:9.2f} it/s  {d['ms_per_step']:.4f} ms/step  p50 {d['gpu_step_ms']['p50']:.4f}  bwd {b[0]['avg_ms']:.4f}  fwd {f[0]['avg_ms']:.4f}  gauss_bwd {g}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for cfg in $CFGS; do
  for t in $THRS; do run c${cfg}_thr$t $cfg FSGS_BWD_MERGE_THR16=$t; done
done 2>&1 | tee $O/summary.txt
