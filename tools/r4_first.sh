#!/bin/bash
# round 4, first GPU call: GPU tests, config-2 sanity line, and a parameter sweep of the surface-bearing dense scene.
O=gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
show() { python tools/show_line.py "$1" "$2"; }
( timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ) 
tail -5 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/c2.json 2> $O/c2.err; show $O/c2.json c2
i=0
for sp in "" "extent=6.0,n_objects=1200,radius=9.0,height=3.5" "scale_lo=0.002,scale_hi=0.015" "extent=6.0,n_objects=1200,radius=9.0,height=3.5,opaque_frac=0.4"; do
  FSGS_SCENE_PARAMS="$sp" timeout 900 python bench.py --config 4 --steps 16 --warmup 8 --no-cpu-baseline > $O/c4_s$i.json 2> $O/c4_s$i.err
  show $O/c4_s$i.json "c4 [$sp]"
  i=$((i+1))
done
timeout 900 python bench.py --config 4 --scene volume --steps 16 --warmup 8 --no-cpu-baseline > $O/c4_vol.json 2> $O/c4_vol.err; show $O/c4_vol.json "c4 volume"
