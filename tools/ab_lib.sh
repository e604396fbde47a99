#!/bin/bash
# A/B of library builds (FSGS_LIB): bash tools/ab_lib.sh TAG "cfgs" name=path [name=path ...]
TAG=$1; CFGS=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O
for r in 1 2; do
  for c in $CFGS; do
    for kv in "$@"; do
      name=${kv%%=*}; lib=${kv#*=}
      FSGS_LIB=$PWD/$lib timeout 400 python bench.py --no-cpu-baseline --no-dropin --config $c > $O/c${c}_${name}_$r.json 2> $O/c${c}_${name}_$r.err
    done
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernels_ms"]
        print(f.split("/")[-1], d["value"], "it/s", d["ms_per_step"], "ms | gpu p50", (d.get("gpu_step_ms") or {}).get("p50"), d.get("live_list_overflows"))
    except Exception as e:
        print(f, "ERR", e)
PY
