#!/bin/bash
O=gpurun_out/r4n; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/profile_refine.py > $O/refine.txt 2>&1; head -70 $O/refine.txt | cut -c1-170
