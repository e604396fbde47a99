set -x
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_express.py tests/test_gpu_occlusion.py -q --durations=5 > gpurun_out/t8.txt 2>&1
tail -12 gpurun_out/t8.txt
timeout 600 python bench.py --no-cpu-baseline --no-dropin > gpurun_out/b8.json 2> gpurun_out/b8.err
python tools/show_line.py gpurun_out/b8.json 2>/dev/null | head -30
python - <<'PY'
import json
d=json.loads(open("gpurun_out/b8.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["iters_per_s_excl_optimizer"], d["gpu_step_ms"], d["host_issue_ms_per_step"], d["roofline"]["valu_roofline"], d["roofline"].get("atomic_roofline"))
PY
