#!/bin/bash
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-dropin "$@" > $O/$name.json 2> $O/$name.err; }
run c2_default --config 2
run c2_h0 --config 2 --handoff 0
run c3_default --config 3
run c3_gate0 --config 3 --handoff-gate-len 0
run c3_gate12 --config 3 --handoff-gate-len 12
run c3_h128 --config 3 --handoff 128
run c3_r3 --config 3 --handoff-rel-len 3
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernels_ms"]; w = d.get("fwd_walk") or {}
        print(f.split("/")[-1], d["value"], "it/s | fwd", k.get("raster_fwd_quad_d4e3", {}).get("avg_ms"), "bwd", k.get("raster_bwd_quad_d4e3", {}).get("avg_ms"), "| tail now", w.get("tail_workgroups_now"), "seen", w.get("handoffs_seen"))
    except Exception as e:
        print(f, "ERR", e)
PY
