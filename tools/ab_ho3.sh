#!/bin/bash
O=gpurun_out/ab_ho3; mkdir -p $O
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-dropin --config 3 "$@" > $O/$name.json 2> $O/$name.err; }
run base1
run h128 --handoff 128
run h256 --handoff 256
run r3 --handoff-rel-len 3
run r6 --handoff-rel-len 6
run q0 --bwd-queue 0
run q4096 --bwd-queue 4096
run base2
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernels_ms"]; w = d.get("fwd_walk") or {}
        print(f.split("/")[-1], d["value"], "it/s", d["ms_per_step"], "ms | gpu p50", (d.get("gpu_step_ms") or {}).get("p50"), "fwd", k.get("raster_fwd_quad_d4e3", {}).get("avg_ms"), "bwd", k.get("raster_bwd_quad_d4e3", {}).get("avg_ms"), "handoffs", w.get("handoffs_seen"))
    except Exception as e:
        print(f, "ERR", e)
PY
