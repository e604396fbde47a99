#!/bin/bash
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O
run() { name=$1; lib=$2; shift 2; FSGS_LIB=$PWD/fusionsense_amd/$lib timeout 300 python bench.py --no-cpu-baseline --no-dropin "$@" > $O/$name.json 2> $O/$name.err; }
for c in 2 3; do
run c${c}_base libfsgs.so --config $c
run c${c}_a_512_8192 libfsgs_sorta.so --config $c
run c${c}_b_512_4096 libfsgs_sortb.so --config $c
run c${c}_c_256_4096 libfsgs_sortc.so --config $c
run c${c}_d_256_2048 libfsgs_sortd.so --config $c
done
FSGS_LIB=$PWD/fusionsense_amd/libfsgs_sortb.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "binning or sort or isect" 2>&1 | tail -2
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernels_ms"]
        print(f.split("/")[-1], d["value"], "it/s | sort", k.get("tile_sort", {}).get("avg_ms"), "count", k.get("isect_count_live", {}).get("avg_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
