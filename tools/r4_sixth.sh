#!/bin/bash
# round 4: the whole GPU suite on the current tree, then the r4 profile sets of configs 2 and 3 (bench line, kernel trace,
# PMC passes), then 4 and 5 on the surface scene
export TMPDIR=/tmp
bash tools/r4_tests.sh r4f
for c in 2 3 4 5; do
  bash tools/collect.sh r4_c$c $c > gpurun_out/collect_c$c.log 2>&1
  python tools/show_line.py gpurun_out/r4_c$c/bench.json "c$c" | head -3
done
