#!/bin/bash
O=gpurun_out/r4af; mkdir -p $O
bash tools/r4_tests.sh r4af
for c in 2 3; do
  timeout 900 python bench.py --config $c --no-cpu-baseline --no-dropin > $O/c${c}.json 2> $O/c${c}.err
  python tools/show_line.py $O/c${c}.json "c$c" | head -2 | cut -c1-260
  python - $O/c${c}.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("   fwd_walk", d.get("fwd_walk"), d.get("fwd_walk_tuning_frames_in_timed_region"), "bwd", d.get("bwd_dispatch"))
PY
done
for c in 4 5; do
  timeout 900 python bench.py --config $c --steps 12 --warmup 4 --no-cpu-baseline > $O/c${c}.json 2> $O/c${c}.err
  python tools/show_line.py $O/c${c}.json "c$c" | head -2 | cut -c1-260
  python - $O/c${c}.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("   fwd_walk", d.get("fwd_walk"), d.get("fwd_walk_tuning_frames_in_timed_region"), "bwd", d.get("bwd_dispatch"))
PY
done
