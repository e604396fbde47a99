#!/bin/bash
# A/B of the compositing backward on the GPU box (through gpurun): bench.py configs #2 and #3 under the switches of
# csrc/raster_live.hip (FSGS_BWD_MERGE_THR16) and a second build with the gradient-line tier 2 off (FSGS_HUGE_LINES=4).
# usage: bash tools/ab_bwd.sh TAG
TAG=${1:-ab}; O=gpurun_out/$TAG; mkdir -p $O
make -C fusionsense_amd/csrc OUT=../libfsgs_h4.so BUILD=build_h4 EXTRA=-DFSGS_HUGE_LINES=4 -j16 > $O/build_h4.log 2>&1
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
    print(f"{sys.argv[2]:28s} {d['value']:9.2f} it/s  {d['ms_per_step']:.4f} ms/step  bwd {b[0]['avg_ms']:.4f}  fwd {f[0]['avg_ms']:.4f}  gauss_bwd {g}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for cfg in 2 3; do
  run c${cfg}_never $cfg FSGS_BWD_MERGE_THR16=17 FSGS_LIB=$PWD/fusionsense_amd/libfsgs_h4.so
  run c${cfg}_never_huge16 $cfg FSGS_BWD_MERGE_THR16=17
  run c${cfg}_thr15 $cfg FSGS_BWD_MERGE_THR16=15
  run c${cfg}_thr14 $cfg FSGS_BWD_MERGE_THR16=14
  run c${cfg}_thr12 $cfg FSGS_BWD_MERGE_THR16=12
  run c${cfg}_always $cfg FSGS_BWD_MERGE_THR16=0
  run c${cfg}_thr14_h4 $cfg FSGS_BWD_MERGE_THR16=14 FSGS_LIB=$PWD/fusionsense_amd/libfsgs_h4.so
done 2>&1 | tee $O/summary.txt
