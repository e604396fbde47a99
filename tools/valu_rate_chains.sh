set -x
O=gpurun_out/valu_rate2; mkdir -p $O tools/build
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/valu_rate.hip -o tools/build/valu_rate || exit 1
timeout 600 tools/build/valu_rate --only-chains > $O/valu_rate.txt 2> $O/valu_rate.err
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $pass | cut -d' ' -f1 | tr 'A-Z' 'a-z')
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_$name -o p -- $ROOT/tools/build/valu_rate --quick --only-chains > $O/pmc_$name.txt 2> $O/pmc_$name.err
done
python tools/valu_rate_pmc.py $O > $O/valu_rate_pmc.txt 2>&1
