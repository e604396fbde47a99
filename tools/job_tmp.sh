export TMPDIR=/tmp
for r in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --no-dropin > gpurun_out/b11_$r.json 2>/dev/null; done
FSGS_BENCH_STEP_EVENTS_EVERY=1 timeout 600 python bench.py --no-cpu-baseline --no-dropin > gpurun_out/b11_e1.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/b11_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["gpu_step_ms"], d["iters_per_s_excl_optimizer"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches_timed"])
PY
timeout 1700 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/gpu_tests_r6d.txt 2>&1
tail -12 gpurun_out/gpu_tests_r6d.txt
