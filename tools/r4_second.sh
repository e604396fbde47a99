#!/bin/bash
# round 4, second GPU call: the whole GPU suite, then configs 4 / 5 on the surface scene (new defaults).
O=gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
show() { python tools/show_line.py "$1" "$2"; }
( timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -15 $O/pytest.log
timeout 900 python bench.py --config 4 --no-cpu-baseline > $O/c4.json 2> $O/c4.err; show $O/c4.json "c4 surface"
timeout 900 python bench.py --config 5 --no-cpu-baseline > $O/c5.json 2> $O/c5.err; show $O/c5.json "c5 surface"
FSGS_OCCLUSION_CUT=0 timeout 900 python bench.py --config 4 --no-cpu-baseline > $O/c4_nocut.json 2> $O/c4_nocut.err; show $O/c4_nocut.json "c4 surface no cuts"
