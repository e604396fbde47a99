set -x
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/ab_lib.sh ab_occ "2" base=fusionsense_amd/libfsgs.so occ8=fusionsense_amd/libfsgs_occ8.so occ7=fusionsense_amd/libfsgs_occ7.so occ5=fusionsense_amd/libfsgs_occ5.so > gpurun_out/ab_occ.txt 2>&1
O=gpurun_out/pmc_wave; mkdir -p $O
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/a -o p -- python3 bench.py --config 2 --no-cpu-baseline --no-dropin > $O/a.json 2> $O/a.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/b -o p -- python3 bench.py --config 2 --no-cpu-baseline --no-dropin > $O/b.json 2> $O/b.err
python - <<'PY' > gpurun_out/pmc_wave/summary.txt 2>&1
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_wave/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    if "fsgs" not in k: continue
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
rm -rf gpurun_out/pmc_wave/a gpurun_out/pmc_wave/b
timeout 1500 python -m pytest tests -m gpu -x -q --durations=60 > gpurun_out/gpu_tests_durations.txt 2>&1
tail -5 gpurun_out/gpu_tests_durations.txt
cat gpurun_out/ab_occ.txt | tail -12
