#!/bin/bash
# bench configs 2 and 3 (and optionally more) once each, print the headline numbers + a few kernel times
TAG=$1; shift; O=gpurun_out/$TAG; mkdir -p $O
for c in ${@:-2 3}; do
  timeout 400 python bench.py --config $c --no-cpu-baseline --no-dropin > $O/c$c.json 2> $O/c$c.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernels_ms"]
        print(f.split("/")[-1], d["value"], "it/s", d["ms_per_step"], "ms | gpu p50", (d.get("gpu_step_ms") or {}).get("p50"),
              {n.replace("raster_","r_").replace("_quad_d4e3",""): v.get("avg_ms") for n, v in k.items()}, d.get("callbacks_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
