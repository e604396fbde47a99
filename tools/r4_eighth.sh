#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "bench_line or eight_ranks or spawns" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); grep -E "^(FAILED|ERROR)|passed|failed|rc=" $O/pytest.log | tail -5
s=$(date +%s); timeout 1200 python bench.py > $O/default.json 2> $O/default.err; e=$(date +%s); echo "default bench wall $((e-s)) s"
python tools/show_line.py $O/default.json default
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4h/default.json').read().strip().splitlines()[-1])
print(json.dumps(d['cpu_baseline'])[:900]); print(d['bwd_dispatch'], d['bwd_dispatch_tuning_frames_in_timed_region'])
PY
