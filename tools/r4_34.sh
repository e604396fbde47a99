#!/bin/bash
mkdir -p gpurun_out/r4ah
( timeout 600 python -m pytest tests/test_gpu_occlusion.py -m gpu -q -k "forward_walks" > gpurun_out/r4ah/pytest.log 2>&1; tail -2 gpurun_out/r4ah/pytest.log )
for c in 3 2; do bash tools/collect.sh r4_c$c $c > gpurun_out/r4ah/collect$c.log 2>&1; python tools/show_line.py gpurun_out/r4_c$c/bench.json "c$c" | cut -c1-260
python - gpurun_out/r4_c$c/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("   fwd_walk", d.get("fwd_walk"), d.get("fwd_walk_tuning_frames_in_timed_region"))
PY
done
