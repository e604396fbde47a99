#!/bin/bash
# N default runs of config #2 under the environment given by the caller; prints the runs with a step over 2 ms.
O=gpurun_out/stall3; mkdir -p $O
for r in $(seq 1 ${1:-40}); do
  timeout 300 python bench.py --no-cpu-baseline --no-dropin > $O/w_$r.json 2> $O/w_$r.err
  python - <<PY
import json
d = json.loads(open("$O/w_$r.json").read().strip().splitlines()[-1])
if d["gpu_step_ms"]["max"] > 2.0:
    print("r$r", d["value"], "p50", d["gpu_step_ms"]["p50"], "max", d["gpu_step_ms"]["max"], d.get("slowest_step"), d.get("slowest_step_view"), d.get("host_ms_around_slowest"))
PY
done
echo done
