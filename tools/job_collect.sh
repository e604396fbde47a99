set -x
export TMPDIR=/tmp
bash tools/collect.sh r6_c2 2 > gpurun_out/collect_r6_c2.log 2>&1
bash tools/collect.sh r6_c3 3 > gpurun_out/collect_r6_c3.log 2>&1
timeout 900 python bench.py --config 2 --steps 300 --densify --no-cpu-baseline --no-dropin > gpurun_out/r6_c2_densify_bench.json 2> gpurun_out/r6_c2_densify_bench.err
bash tools/step_trace.sh 2 > gpurun_out/r6_step_trace.txt 2>&1
ls -la gpurun_out/r6_c2 gpurun_out/r6_c3
python tools/show_bench.py gpurun_out/r6_c2/bench.json | tail -15
python tools/show_bench.py gpurun_out/r6_c2_densify_bench.json | grep -E "value|binning|overflow|express|callbacks|ms_per_step" 
