#!/bin/bash
# The launches of the slowest and of the median step of one bench.py run (rocprofv3 --kernel-trace): tools/step_trace.sh CFG
O=gpurun_out/steptrace; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/t
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 bench.py --no-cpu-baseline --no-dropin --config ${1:-2} > $O/bench.json 2> $O/bench.err
python tools/stall_trace.py $(find $O/t -name "*kernel_trace.csv") ${2:-1} 60
rm -rf $O/t
