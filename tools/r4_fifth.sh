#!/bin/bash
# round 4: wave-tier tile sort A/B + the binning tests that pin it bit for bit + host profile of the shim route
O=gpurun_out/r4e; mkdir -p $O
export TMPDIR=/tmp
show() { python tools/show_line.py "$1" "$2" | head -2; }
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_occlusion.py -m gpu -q -x -k "binning or sort or bin or lists or fused or fullsize or config2 or dense" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $O/pytest.log | tail -8
for w in 1 0; do
  FSGS_SORT_WAVE_TIER=$w timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-dropin > $O/c2_wave$w.json 2> $O/c2_wave$w.err; show $O/c2_wave$w.json "c2 wave tier $w"
  FSGS_SORT_WAVE_TIER=$w timeout 600 python bench.py --config 3 --no-cpu-baseline --no-dropin > $O/c3_wave$w.json 2> $O/c3_wave$w.err; show $O/c3_wave$w.json "c3 wave tier $w"
done
timeout 600 python tools/profile_dropin.py > $O/profile_dropin.txt 2>&1; head -60 $O/profile_dropin.txt
( timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_integration.py -m gpu -q -k "cosine or other_loss or pseudo" > $O/pytest2.log 2>&1; echo "pytest rc=$?" >> $O/pytest2.log ); grep -E "^(FAILED|ERROR)|passed|failed|rc=" $O/pytest2.log | tail -6
for c in 2 3 4; do FSGS_LIB=$PWD/fusionsense_amd/libfsgs_stats.so timeout 600 python tools/bwd_lane_stats.py $c 2> $O/lanes_c$c.err | tail -1 | tee $O/lanes_c$c.json; done
