#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python - > $O/knn.log 2>&1 <<'PY'
import time, torch
from fusionsense_amd.inference import knn_drop_first
dev=torch.device("cuda:0")
g=torch.Generator().manual_seed(0)
x=torch.randn(300_000,3,generator=g).to(dev)
for m in (20_000, 200_000, 640_000):
    y=torch.randn(m,3,generator=g).to(dev)
    torch.cuda.synchronize(); t=time.perf_counter(); out=knn_drop_first(x,y,16); torch.cuda.synchronize()
    print("knn", m, "x 300k, k=16:", round((time.perf_counter()-t)*1e3,1), "ms", flush=True)
PY
echo "knn rc=$?"; tail -5 $O/knn.log
timeout 600 python - > $O/ls.log 2>&1 <<'PY'
import time, torch
dev=torch.device("cuda:0")
from fusionsense_amd import scenes, inference as inf
from fusionsense_amd.trainer import SplatTrainer
tr=SplatTrainer(scenes.lego_like_scene(300_000, seed=0), dev, sh_degree=3)
cam=scenes.hemisphere_cameras(1)[0]
for it in range(2):
    torch.cuda.synchronize(); t=time.perf_counter(); r=inf.compute_level_surface_points(tr, cam, 100000); torch.cuda.synchronize()
    print("level sets of one 800x800 frame:", round((time.perf_counter()-t)*1e3,1), "ms", {k: v["points"].shape[0] for k,v in r.items()}, flush=True)
PY
echo "ls rc=$?"; tail -8 $O/ls.log
