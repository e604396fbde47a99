#!/bin/bash
O=gpurun_out/ab_c5; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-dropin --config 5 > $O/$name.json 2> $O/$name.err; }
run base A=1
run noskip FSGS_SKIP_IDLE_MIN_N=1000000000
run nolazy FSGS_LAZY_SH_MIN_N=1000000000
run neither FSGS_SKIP_IDLE_MIN_N=1000000000 FSGS_LAZY_SH_MIN_N=1000000000
run base2 A=1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernels_ms"]
        print(f.split("/")[-1], d["value"], "it/s", d["ms_per_step"], "ms | gauss", k.get("gaussian_bwd",{}).get("avg_ms"), "never_updated", d.get("never_updated_gaussians_frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
