#!/bin/bash
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-dropin "$@" > $O/$name.json 2> $O/$name.err; }
run c3_q0 --config 3 --bwd-queue 0
run c3_q1024 --config 3 --bwd-queue 1024
run c3_q4096 --config 3 --bwd-queue 4096
run c2_q0 --config 2 --bwd-queue 0
run c2_q1024 --config 2 --bwd-queue 1024
run c3_r5 --config 3 --handoff-rel-len 5
run c3_r6 --config 3 --handoff-rel-len 6
run c2_r6 --config 2 --handoff-rel-len 6
run c4_q0 --config 4 --bwd-queue 0
run c4_q1024 --config 4 --bwd-queue 1024
run c4_q8192 --config 4 --bwd-queue 8192
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernels_ms"]; w = d.get("fwd_walk") or {}
        print(f.split("/")[-1], d["value"], "it/s | fwd", k.get("raster_fwd_quad_d4e3", {}).get("avg_ms"), "bwd", k.get("raster_bwd_quad_d4e3", {}).get("avg_ms"), "| tail now", w.get("tail_workgroups_now"), "seen", w.get("handoffs_seen"))
    except Exception as e:
        print(f, "ERR", e)
PY
