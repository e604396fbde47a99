#!/bin/bash
# rocprofv3 kernel table of one bench run: bash tools/prof_quick.sh CONFIG [pattern]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=$1; PAT=${2:-.}
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/pq_$C -o t -- python3 bench.py --config $C --no-cpu-baseline --no-dropin > gpurun_out/pq_$C.json 2> gpurun_out/pq_$C.err
python tools/rocpd_summary.py gpurun_out/pq_$C/t_results.db | grep -E "$PAT" | awk -F'",' '{n=split($2,a,","); printf "%-70s %s\n", substr($1,2,70), $2}' | head -16
rm -rf gpurun_out/pq_$C
