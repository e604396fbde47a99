#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q -k "level_set or level_sets or knn" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); grep -E "^(FAILED|ERROR)|^E  |passed|failed|rc=" $O/pytest.log | tail -12
timeout 600 python - > $O/ls.log 2>&1 <<'PY'
import time, torch
dev=torch.device("cuda:0")
from fusionsense_amd import scenes, inference as inf
from fusionsense_amd.trainer import SplatTrainer
tr=SplatTrainer(scenes.lego_like_scene(300_000, seed=0), dev, sh_degree=3)
cam=scenes.hemisphere_cameras(1)[0]
for it in range(3):
    torch.cuda.synchronize(); t=time.perf_counter(); r=inf.compute_level_surface_points(tr, cam, 100000); torch.cuda.synchronize()
    print("level sets of one 800x800 frame:", round((time.perf_counter()-t)*1e3,1), "ms", {k: v["points"].shape[0] for k,v in r.items()}, flush=True)
PY
echo "ls rc=$?"; tail -4 $O/ls.log
