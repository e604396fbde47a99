#!/bin/bash
O=gpurun_out/r4ai; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_occlusion.py -m gpu -q -x -k "fused or walk or cut" > $O/pytest.log 2>&1; tail -2 $O/pytest.log )
for c in 2 4; do
  st="--steps 40 --warmup 10"; [ $c -ge 4 ] && st="--steps 12 --warmup 4"
  FSGS_FWD_WALK=1 timeout 900 python bench.py --config $c $st --no-cpu-baseline --no-dropin > $O/c${c}.json 2> $O/c${c}.err
  python tools/show_line.py $O/c${c}.json "c$c" | head -2 | cut -c1-200
done
