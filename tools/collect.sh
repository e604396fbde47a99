#!/bin/bash
# Collect one configuration's measurement artefacts on the GPU box into gpurun_out/<tag>/ :
#   bench line, rocprofv3 --kernel-trace --stats table, and the PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters; each in
#   its own run, as MI355X_MICROARCH.md prescribes), summarised by tools/pmc_summary.py.
# usage (through gpurun):  bash tools/collect.sh r2_c2 2 [extra bench.py arguments for the profiled runs]
set -x
TAG=$1; CFG=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && cd $ROOT
timeout 900 python bench.py --config $CFG > $O/bench.json 2> $O/bench.err
# (round 5: the forward's walk is fixed — nothing to carry over from the plain run to the profiled ones)
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 bench.py --config $CFG --no-cpu-baseline --no-dropin "$@" > $O/trace_bench.json 2> $O/trace.err
python tools/rocpd_summary.py $O/trace/t_results.db > $O/kernel_stats.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  name=$(echo $pass | cut -d' ' -f1 | tr 'A-Z' 'a-z')
  timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_$name -o p -- python3 bench.py --config $CFG --no-cpu-baseline --no-dropin "$@" > $O/pmc_$name.json 2> $O/pmc_$name.err
done
python tools/pmc_summary.py $O/pmc_hbm.csv $O/pmc_traffic.json $CFG "$TAG" $(find $O/pmc_fetch_size $O/pmc_write_size $O/pmc_sq_insts_valu -name '*counter_collection.csv')
rm -rf $O/trace/t_results.db.tmp $O/trace/*.db $O/pmc_*/ 2>/dev/null
ls -la $O
