set -x
export TMPDIR=/tmp
bash tools/collect.sh r6_c4 4 > gpurun_out/collect_r6_c4.log 2>&1
bash tools/collect.sh r6_c5 5 > gpurun_out/collect_r6_c5.log 2>&1
timeout 900 python bench.py --config 2 --steps 300 --densify --no-cpu-baseline --no-dropin > gpurun_out/r6_c2_densify_bench.json 2> gpurun_out/r6_c2_densify_bench.err
ls -la gpurun_out/r6_c4 gpurun_out/r6_c5
python tools/show_bench.py gpurun_out/r6_c4/bench.json | tail -12
python tools/show_bench.py gpurun_out/r6_c5/bench.json | tail -12
