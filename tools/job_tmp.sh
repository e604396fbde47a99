bash tools/collect.sh r6b_c3 3 > gpurun_out/collect_c3.log 2>&1
python bench.py --config 2 --steps 300 --densify --no-cpu-baseline --no-dropin > gpurun_out/c2_densify.json 2> gpurun_out/c2_densify.err
python - <<EOP
import json
d=json.loads(open("gpurun_out/r6b_c3/bench.json").read().strip().splitlines()[-1])
print("c3", d["value"], d["callbacks_ms"], d["iters_per_s_excl_callbacks"], d["max_step_ms"], d["device_mallocs_in_timed_region"], d["gpu_step_ms"], d.get("dropin_iters_per_s"), d.get("patched_full_iters_per_s"))
d=json.loads(open("gpurun_out/c2_densify.json").read().strip().splitlines()[-1])
print("c2 densify", d["value"], d["ms_per_step"], d.get("callbacks_ms"), d["gpu_step_ms"], d["binning_frames"], d["live_list_overflows"], d["config"].get("refinement"), d["max_step_ms"])
EOP
