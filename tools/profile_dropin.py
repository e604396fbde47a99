"""Host-side profile of the two-line-shim route (rasterization() + rasterize_gaussians() under the reference-style
op-by-op caller): cProfile over 60 training steps, top functions by cumulative and by own time."""
import cProfile
import io
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusionsense_amd import scenes  # noqa: E402
from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig  # noqa: E402
from fusionsense_amd.trainer import SplatTrainer  # noqa: E402

dev = torch.device("cuda:0")
N, R = 300_000, 800
cams = scenes.hemisphere_cameras(8, width=R, height=R, focal=1111.11, seed=0)
params = scenes.lego_like_scene(N, seed=0)
tgt_tr = SplatTrainer(scenes.lego_like_scene(N, seed=1), dev, sh_degree=3)
targets = []
with torch.no_grad():
    for c in cams:
        o = tgt_tr.forward(c)
        targets.append({k: o[k].detach().clone() for k in ("rgb", "depth", "normal")})
del tgt_tr
st = DensifyStrategy(SplatfactoConfig(), num_train_data=8, stats_only=True)
tr = SplatTrainer(params, dev, sh_degree=3, fused=False, strategy=st)
for s in range(10):
    tr.train_step(cams[s % 8], targets[s % 8])
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for s in range(30):
    tr.train_step(cams[s % 8], targets[s % 8])
torch.cuda.synchronize()
print(f"drop-in route: {30 / (time.perf_counter() - t0):.1f} it/s")
pr = cProfile.Profile()
pr.enable()
for s in range(60):
    tr.train_step(cams[s % 8], targets[s % 8])
torch.cuda.synchronize()
pr.disable()
for key in ("cumulative", "tottime"):
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(45)
    print(buf.getvalue()[:9000])
