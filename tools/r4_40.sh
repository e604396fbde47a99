#!/bin/bash
mkdir -p gpurun_out/r4an
bash tools/r4_tests.sh r4an
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4an/smoke.log 2>&1; tail -2 gpurun_out/r4an/smoke.log
bash tools/collect.sh r4_c2 2 > gpurun_out/r4an/collect2.log 2>&1; python tools/show_line.py gpurun_out/r4_c2/bench.json "c2" | cut -c1-260
( time python bench.py > gpurun_out/r4an/default_bench.json 2> gpurun_out/r4an/default_bench.err ) 2>&1 | grep real
python tools/show_line.py gpurun_out/r4an/default_bench.json "default" | cut -c1-200
