#!/bin/bash
# A/B of one environment variable on one box: bash tools/ab_env.sh TAG "cfgs" VAR "values" [runs]
TAG=$1; CFGS=$2; VAR=$3; VALS=$4; RUNS=${5:-2}
O=gpurun_out/$TAG; mkdir -p $O
for r in $(seq 1 $RUNS); do
  for c in $CFGS; do
    for v in $VALS; do
      env $VAR=$v timeout 600 python bench.py --no-cpu-baseline --no-dropin --config $c > $O/c${c}_v${v}_$r.json 2> $O/c${c}_v${v}_$r.err
    done
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernels_ms"]
        print(f.split("/")[-1], d["value"], "it/s", d["ms_per_step"], "ms | gpu p50", (d.get("gpu_step_ms") or {}).get("p50"), "gauss", k.get("gaussian_bwd", {}).get("avg_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
