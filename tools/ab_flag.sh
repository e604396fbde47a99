#!/bin/bash
# A/B of one bench.py flag on one box: bash tools/ab_flag.sh TAG "cfgs" --flag valA valB [runs]
TAG=$1; CFGS=$2; FLAG=$3; A=$4; B=$5; RUNS=${6:-2}
O=gpurun_out/$TAG; mkdir -p $O
for r in $(seq 1 $RUNS); do
  for c in $CFGS; do
    for v in $A $B; do
      timeout 400 python bench.py --no-cpu-baseline --no-dropin --config $c $FLAG $v > $O/c${c}_v${v}_$r.json 2> $O/c${c}_v${v}_$r.err
    done
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernels_ms"]
        print(f.split("/")[-1], d["value"], "it/s", d["ms_per_step"], "ms | gpu p50", (d.get("gpu_step_ms") or {}).get("p50"), "bwd", k.get("raster_bwd_quad_d4e3", {}).get("avg_ms"), "fwd", k.get("raster_fwd_quad_d4e3", {}).get("avg_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
