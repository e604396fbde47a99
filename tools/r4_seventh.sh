#!/bin/bash
O=gpurun_out/r4g; mkdir -p $O
export TMPDIR=/tmp
for c in 2 3 4; do FSGS_LIB=$PWD/fusionsense_amd/libfsgs_stats.so timeout 600 python tools/bwd_lane_stats.py $c 2> $O/lanes_c$c.err | tail -1 | tee $O/lanes_c$c.json; done
timeout 900 python bench.py --steps 20 --warmup 5 > $O/c2_driver.json 2> $O/c2_driver.err; python tools/show_line.py $O/c2_driver.json "c2 driver-style"
timeout 900 python bench.py --config 3 --no-cpu-baseline > $O/c3.json 2> $O/c3.err; python tools/show_line.py $O/c3.json "c3"
