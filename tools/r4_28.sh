#!/bin/bash
O=gpurun_out/r4ab; mkdir -p $O
export TMPDIR=/tmp
bash tools/r4_tests.sh r4ab
for c in 2 3 4 5; do for l in libfsgs_gen2.so libfsgs.so; do
  st="--steps 40 --warmup 10"; [ $c -ge 4 ] && st="--steps 12 --warmup 4"
  FSGS_LIB=$PWD/fusionsense_amd/$l timeout 900 python bench.py --config $c $st --no-cpu-baseline --no-dropin > $O/c${c}_$l.json 2> $O/c${c}_$l.err
  python tools/show_line.py $O/c${c}_$l.json "c$c $l" | head -2 | cut -c1-200
done; done
