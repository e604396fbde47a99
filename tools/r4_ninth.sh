#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
export TMPDIR=/tmp
for c in 4 2 3; do timeout 900 python tools/grad_coverage.py $c 2> $O/cov_c$c.err | tail -1 | tee $O/cov_c$c.json | cut -c1-1500; done
