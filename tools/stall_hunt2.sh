#!/bin/bash
# The rare stall of config #2's default run under rocprofv3 --kernel-trace: is it a kernel or a gap?
O=gpurun_out/stall2; mkdir -p $O; export TMPDIR=/tmp
for r in $(seq 1 ${1:-8}); do
  rm -rf $O/t$r
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t$r -o t -- python3 bench.py --no-cpu-baseline --no-dropin > $O/r$r.json 2> $O/r$r.err
  python - <<PY
import json
d = json.loads(open("$O/r$r.json").read().strip().splitlines()[-1])
print("r$r", d["value"], d["gpu_step_ms"]["max"], d.get("slowest_step"), d.get("host_ms_around_slowest"))
PY
  python tools/stall_trace.py $(find $O/t$r -name "*kernel_trace.csv") ${2:-2} 60
  rm -rf $O/t$r
done
