#!/bin/bash
O=gpurun_out/r4k; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "knn or level_set" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); grep -E "^(FAILED|ERROR)|^E  |passed|failed|rc=" $O/pytest.log | tail -12
python - <<'PY'
import time, torch
from fusionsense_amd.inference import knn_drop_first
dev=torch.device("cuda:0")
g=torch.Generator().manual_seed(0)
x=torch.randn(300_000,3,generator=g).to(dev); y=torch.randn(640_000,3,generator=g).to(dev)
knn_drop_first(x[:1000], y[:1000], 16); torch.cuda.synchronize()
t=time.perf_counter(); out=knn_drop_first(x,y,16); torch.cuda.synchronize(); print("knn 640k x 300k, k=16:", round((time.perf_counter()-t)*1e3,1), "ms")
# level-set extraction of one 800x800 frame of config #2's scene
from fusionsense_amd import scenes, inference as inf
from fusionsense_amd.trainer import SplatTrainer
tr=SplatTrainer(scenes.lego_like_scene(300_000, seed=0), dev, sh_degree=3)
cam=scenes.hemisphere_cameras(1)[0]
inf.compute_level_surface_points(tr, cam, 100000); torch.cuda.synchronize()
t=time.perf_counter(); r=inf.compute_level_surface_points(tr, cam, 100000); torch.cuda.synchronize()
print("level sets of one 800x800 frame:", round((time.perf_counter()-t)*1e3,1), "ms", {k: v["points"].shape[0] for k,v in r.items()})
PY
