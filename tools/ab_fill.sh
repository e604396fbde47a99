#!/bin/bash
# A/B of the count-free binning route (fused.BIN_FILL) against the two-pass route, alternating runs on one box.
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-dropin "$@" > $O/$name.json 2> $O/$name.err; }
for r in 1 2 3; do
  run c2_fill_$r --config 2 --bin-fill 1
  run c2_two_$r --config 2 --bin-fill 0
  run c3_fill_$r --config 3 --bin-fill 1
  run c3_two_$r --config 3 --bin-fill 0
done
run c4_fill --config 4 --bin-fill 1
run c4_two --config 4 --bin-fill 0
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], "it/s", d["ms_per_step"], "ms | gpu p50", (d.get("gpu_step_ms") or {}).get("p50"), d.get("binning_frames"), "overflows", d.get("live_list_overflows"), "N", d.get("config", {}).get("n_gaussians_end"))
    except Exception as e:
        print(f, "ERR", e)
PY
