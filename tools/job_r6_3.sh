set -x
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_express.py -x -q --durations=8 > gpurun_out/express_tests.txt 2>&1
tail -25 gpurun_out/express_tests.txt
bash tools/ab_flag.sh ab_express "2 3" --express 0 1 2 > gpurun_out/ab_express.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_express/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d.get("express_steps_timed"), d.get("host_issue_ms_per_step"), d.get("gpu_step_ms"), d["roofline"].get("valu_roofline",{}).get("frac"), d.get("live_list_overflows"), d.get("binning_frames"))
    except Exception as e: print(f, "ERR", e)
PY
bash tools/ab_lib.sh ab_pf "2" base=fusionsense_amd/libfsgs.so pf1o5=fusionsense_amd/libfsgs_pf1o5.so pf2o5=fusionsense_amd/libfsgs_pf2o5.so > gpurun_out/ab_pf.txt 2>&1
tail -8 gpurun_out/ab_pf.txt
