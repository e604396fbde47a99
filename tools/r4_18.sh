#!/bin/bash
bash tools/r4_tests.sh r4r
bash tools/collect.sh r4_c4 4 > gpurun_out/r4r/collect4.log 2>&1
bash tools/collect.sh r4_c5 5 > gpurun_out/r4r/collect5.log 2>&1
python tools/show_line.py gpurun_out/r4_c4/bench.json gpurun_out/r4_c5/bench.json
