#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of one bench configuration: bash tools/prof_fwd.sh <tag> <config> [bench args]
TAG=$1; CFG=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && cd $ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 bench.py --config $CFG --no-cpu-baseline --no-dropin "$@" > $O/trace_bench.json 2> $O/trace.err
python tools/rocpd_summary.py $O/trace/t_results.db > $O/kernel_stats.csv
rm -rf $O/trace
head -16 $O/kernel_stats.csv | cut -c1-200
