#!/bin/bash
# the round's closing check: whole GPU suite, smoke, the driver's default bench call
mkdir -p gpurun_out/r4_final
bash tools/r4_tests.sh r4_final
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_final/smoke.log 2>&1; tail -1 gpurun_out/r4_final/smoke.log
( time python bench.py > gpurun_out/r4_final/default_bench.json 2> gpurun_out/r4_final/default_bench.err ) 2>&1 | grep real
python tools/show_line.py gpurun_out/r4_final/default_bench.json "default" | cut -c1-200
