#!/bin/bash
# Calibration run of tools/valu_rate.hip on the GPU box (through gpurun):  bash tools/valu_rate.sh
#   1. plain run: cycles per wave64 instruction per class / EXEC mask / waves per SIMD, atomics' rates -> valu_rate.txt
#   2. rocprofv3 --pmc passes over the chip-wide launches (each counter set in its own run, no trace domains besides
#      --kernel-trace): SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES / SQ_WAVE_CYCLES per chain kernel (what unit
#      SQ_ACTIVE_INST_VALU counts in), WRITE_SIZE and FETCH_SIZE per atomic / store pattern (bytes the counters charge per
#      global_atomic_add_f32).  Raw CSVs stay under gpurun_out/valu_rate/; tools/valu_rate_pmc.py tabulates them.
set -x
O=gpurun_out/valu_rate; mkdir -p $O tools/build
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && cd $ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/valu_rate.hip -o tools/build/valu_rate || exit 1
timeout 600 tools/build/valu_rate > $O/valu_rate.txt 2> $O/valu_rate.err
rocprofv3 -L > $O/counters_available.txt 2>&1
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $pass | cut -d' ' -f1 | tr 'A-Z' 'a-z')
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_$name -o p -- $ROOT/tools/build/valu_rate --quick --only-chains > $O/pmc_$name.txt 2> $O/pmc_$name.err
done
for pass in "WRITE_SIZE" "FETCH_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  name=$(echo $pass | cut -d' ' -f1 | tr 'A-Z' 'a-z')
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_$name -o p -- $ROOT/tools/build/valu_rate --only-atoms > $O/pmc_$name.txt 2> $O/pmc_$name.err
done
find $O -name '*counter_collection.csv' | xargs ls -la
python tools/valu_rate_pmc.py $O > $O/valu_rate_pmc.txt 2>&1
tail -n 80 $O/valu_rate.txt
