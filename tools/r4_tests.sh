#!/bin/bash
# the whole GPU suite (+ optional extra pytest args), log under gpurun_out/$1
O=gpurun_out/${1:-tests}; mkdir -p $O; shift
export TMPDIR=/tmp
( timeout 2700 python -m pytest tests -m gpu -q "$@" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $O/pytest.log | tail -20
