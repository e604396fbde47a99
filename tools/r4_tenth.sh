#!/bin/bash
O=gpurun_out/r4j; mkdir -p $O
export TMPDIR=/tmp
bash tools/r4_tests.sh r4j
timeout 900 python bench.py --config 3 --start-step 0 --steps 15000 --warmup 0 --no-cpu-baseline --no-dropin > $O/c3_full.json 2> $O/c3_full.err; python tools/show_line.py $O/c3_full.json "c3 whole schedule" | head -2
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4j/c3_full.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','iters_per_s_excl_callbacks','live_list_overflows','device_mallocs_in_timed_region','gpu_step_ms')}, d['config'].get('refinement'), d['config'].get('n_gaussians'))
PY
