#!/bin/bash
O=gpurun_out/ab_ho; mkdir -p $O
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-dropin "$@" > $O/$name.json 2> $O/$name.err; }
for r in 1 2; do
run c2_base_$r --config 2
run c2_g0_r4_$r --config 2 --handoff-gate-len 0 --handoff-rel-len 4
run c2_g0_r3_$r --config 2 --handoff-gate-len 0 --handoff-rel-len 3
run c2_g0_r3_h128_$r --config 2 --handoff-gate-len 0 --handoff-rel-len 3 --handoff 128
done
run c4_order1 --config 4 --sort-order 1
run c4_order0 --config 4 --sort-order 0
run c4_order1b --config 4 --sort-order 1
run c4_order0b --config 4 --sort-order 0
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernels_ms"]; w = d.get("fwd_walk") or {}
        print(f.split("/")[-1], d["value"], "it/s", d["ms_per_step"], "ms | gpu p50", (d.get("gpu_step_ms") or {}).get("p50"), "fwd", k.get("raster_fwd_quad_d4e3", {}).get("avg_ms"), "handoffs", w.get("handoffs_seen"), "tail", w.get("tail_workgroups_now"))
    except Exception as e:
        print(f, "ERR", e)
PY
