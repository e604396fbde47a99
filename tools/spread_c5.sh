#!/bin/bash
# Run-to-run spread of config #5 on ONE box: N back-to-back default runs, it/s and the dominant launch's
# time of each -> gpurun_out/spread_c5.txt    usage: bash tools/spread_c5.sh [N]
N=${1:-8}; O=gpurun_out/spread_c5; mkdir -p $O; export TMPDIR=/tmp
for r in $(seq 1 $N); do
  timeout 900 python bench.py --config 5 --no-cpu-baseline --no-dropin > $O/run_$r.json 2> $O/run_$r.err
done
python - <<PY > gpurun_out/spread_c5.txt
import json, glob, statistics
rows = []
for f in sorted(glob.glob("$O/run_*.json"), key=lambda p: int(p.split("_")[-1].split(".")[0])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernels_ms"]
        rows.append((d["value"], d["ms_per_step"], k["gaussian_bwd"]["avg_ms"], k["raster_bwd_quad_d4e3"]["avg_ms"], k["isect_count_live"]["avg_ms"], d["gpu_step_ms"]["p50"]))
        print(f.split("/")[-1], *rows[-1])
    except Exception as e:
        print(f, "ERR", e)
if rows:
    v = [r[0] for r in rows]; g = [r[2] for r in rows]
    print("it/s  min %.1f median %.1f max %.1f spread %.1f %%" % (min(v), statistics.median(v), max(v), 100 * (max(v) - min(v)) / statistics.median(v)))
    print("gauss_sh_bwd<HALF,ADAM> ms  min %.3f median %.3f max %.3f" % (min(g), statistics.median(g), max(g)))
PY
cat gpurun_out/spread_c5.txt
