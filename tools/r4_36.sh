#!/bin/bash
O=gpurun_out/r4aj; mkdir -p $O
for l in libfsgs.so libfsgs_emp.so libfsgs.so libfsgs_emp.so; do
  FSGS_LIB=$PWD/fusionsense_amd/$l FSGS_FWD_WALK=1 timeout 900 python bench.py --config 2 --steps 40 --warmup 10 --no-cpu-baseline --no-dropin > $O/c2_$l.json 2> $O/c2_$l.err
  python tools/show_line.py $O/c2_$l.json "c2 $l" | head -2 | tail -1 | cut -c1-200
done
