set -x
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_express.py -x -q --durations=8 > gpurun_out/express_tests.txt 2>&1
tail -25 gpurun_out/express_tests.txt
bash tools/ab_flag.sh ab_express "2" --express 0 1 3 > gpurun_out/ab_express.txt 2>&1
tail -8 gpurun_out/ab_express.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_express/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d.get("express_steps_timed"), d.get("host_issue_ms_per_step"), d.get("gpu_step_ms"), d["roofline"].get("valu_roofline",{}).get("frac"), d["roofline"].get("atomic_roofline"))
    except Exception as e: print(f, "ERR", e)
PY
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_dense.py -x -q --durations=12 -k "oracle" > gpurun_out/oracle_tests.txt 2>&1
tail -20 gpurun_out/oracle_tests.txt
