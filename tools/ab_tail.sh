#!/bin/bash
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-dropin "$@" > $O/$name.json 2> $O/$name.err; }
run c2_a_plain --config 2 --handoff 0
for r in 4 8; do for h in 64 128 192; do
run c2_h${h}_r$r --config 2 --handoff $h --handoff-rel-len $r --tail-items 4096
run c3_h${h}_r$r --config 3 --handoff $h --handoff-rel-len $r --tail-items 8192
done; done
run c3_h128_r8_q2048 --config 3 --handoff 128 --handoff-rel-len 8 --tail-items 2048
run c3_four --config 3 --fwd-walk 0
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernels_ms"]
        print(f.split("/")[-1], d["value"], "it/s | fwd", k.get("raster_fwd_quad_d4e3", {}).get("avg_ms"), (d.get("fwd_walk") or {}).get("tail_error"))
    except Exception as e:
        print(f, "ERR", e)
PY
