#!/bin/bash
# round 4, third GPU call: A/B of the per-Gaussian backward + Adam (batched geometry Adam, one-round-trip feature Adam).
O=gpurun_out/r4c; mkdir -p $O
export TMPDIR=/tmp
show() { python tools/show_line.py "$1" "$2" | head -2; }
( timeout 900 python -m pytest tests/test_gpu_occlusion.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "adam or dispatch or level_set or unlisted or kernels_of or c_abi or gauss or eight" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -4 $O/pytest.log
for l in libfsgs.so libfsgs_u4.so libfsgs_u8.so; do
  FSGS_LIB=$PWD/fusionsense_amd/$l timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-dropin > $O/c2_$l.json 2> $O/c2_$l.err; show $O/c2_$l.json "c2 $l"
done
FSGS_SKIP_IDLE_MIN_N=0 timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-dropin > $O/c2_skip.json 2> $O/c2_skip.err; show $O/c2_skip.json "c2 skip-idle forced"
timeout 600 python bench.py --config 3 --no-cpu-baseline --no-dropin > $O/c3.json 2> $O/c3.err; show $O/c3.json "c3"
for l in libfsgs.so libfsgs_u8.so; do
  FSGS_LIB=$PWD/fusionsense_amd/$l timeout 900 python bench.py --config 4 --no-cpu-baseline > $O/c4_$l.json 2> $O/c4_$l.err; show $O/c4_$l.json "c4 $l"
done
FSGS_SKIP_IDLE_MIN_N=100000000 FSGS_LAZY_SH_MIN_N=100000000 timeout 900 python bench.py --config 4 --no-cpu-baseline > $O/c4_noskip.json 2> $O/c4_noskip.err; show $O/c4_noskip.json "c4 no skip / no lazy"
FSGS_SKIP_IDLE_MIN_N=100000000 timeout 900 python bench.py --config 4 --no-cpu-baseline > $O/c4_noskip_lazy.json 2> $O/c4_noskip_lazy.err; show $O/c4_noskip_lazy.json "c4 no skip, lazy"
