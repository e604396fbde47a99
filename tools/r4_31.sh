#!/bin/bash
bash tools/r4_tests.sh r4ae
for c in 2 3 4 5; do bash tools/collect.sh r4_c$c $c > gpurun_out/r4ae/collect$c.log 2>&1; python tools/show_line.py gpurun_out/r4_c$c/bench.json "c$c" | cut -c1-260; done
