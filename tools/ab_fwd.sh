#!/bin/bash
# A/B of the forward compositing's walk on the GPU box: bench.py per configuration with the four-wave walk, the plain
# one-wave walk and the hand-off at several thresholds.  usage (through gpurun): bash tools/ab_fwd.sh <tag> "<configs>" "<handoffs>"
TAG=$1; CFGS=${2:-"2 3"}; HS=${3:-"0 64 128 192"}
O=gpurun_out/$TAG; mkdir -p $O
for c in $CFGS; do
  timeout 300 python bench.py --config $c --no-cpu-baseline --no-dropin --fwd-walk 0 > $O/c${c}_four.json 2> $O/c${c}_four.err
  for h in $HS; do
    timeout 300 python bench.py --config $c --no-cpu-baseline --no-dropin --handoff $h > $O/c${c}_h$h.json 2> $O/c${c}_h$h.err
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d["kernels_ms"]
        print(f.split("/")[-1], d["value"], "it/s", d["ms_per_step"], "ms | fwd", k.get("raster_fwd_quad_d4e3", {}).get("avg_ms"),
              "bwd", k.get("raster_bwd_quad_d4e3", {}).get("avg_ms"), "| walk", d.get("fwd_walk"))
    except Exception as e:
        print(f, "ERR", e)
PY
