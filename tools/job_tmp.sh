export TMPDIR=/tmp
for cfg in "0.10 1.25" "0.10 1.5" "0.10 2.0" "0.02 1.25"; do
  set -- $cfg
  FSGS_BIN_HIST_KEEP_REL=$1 FSGS_BIN_FILL_GROWTH=$2 timeout 900 python bench.py --config 2 --steps 300 --densify --no-cpu-baseline --no-dropin > gpurun_out/dens_$1_$2.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/dens_$1_$2.json').read().strip().splitlines()[-1])
print("$1 $2", d["value"], d["live_list_overflows"], d["binning_frames"], d["gpu_step_ms"]["p50"], d["gpu_step_ms"]["mean_by_quarter"], d["gpu_step_ms"]["max"])
PY
done
FSGS_BIN_FILL_GROWTH=1.5 timeout 600 python bench.py --no-cpu-baseline --no-dropin > gpurun_out/c2_g15.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/c2_g15.json').read().strip().splitlines()[-1])
print("plain growth 1.5", d["value"], d["gpu_step_ms"]["p50"], {k:v["avg_ms"] for k,v in d["kernels_ms"].items() if k in ("isect_count_live","tile_sort")})
PY
