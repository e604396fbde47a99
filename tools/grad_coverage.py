"""Per-frame gradient coverage: the share of Gaussians whose SH features / geometry receive a non-zero gradient in ONE
frame (what a lazily caught-up Adam would have to touch), per configuration.
    python tools/grad_coverage.py [config]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sys.argv = ["bench.py", "--config", str(cfg), "--no-cpu-baseline", "--no-dropin"] + (["--views", "8"] if cfg == 2 else [])
args = bench.parse()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
wl = bench.build_workload(args, dev)
tr, cams, targets = wl["trainer"], wl["cams"], wl["targets"]
for s in range(len(cams)):
    tr.train_step(cams[s], targets[s])
rows = []
for s in range(len(cams)):
    _, out = tr.train_step(cams[s], targets[s], optimizer_step=False)  # (gradients land in the slab)
    g = tr.gradients()
    n = tr.num_gaussians()
    info = out["info"]
    rows.append({"view": s,
                 "features_dc": round(float((g["features_dc"].abs().sum(-1) > 0).float().mean()), 4),
                 "means": round(float((g["means"].abs().sum(-1) > 0).float().mean()), 4),
                 "opacities": round(float((g["opacities"].abs().reshape(n) > 0).float().mean()), 4),
                 "visible": round(float((info.radii[0] > 0).float().mean()), 4),
                 "listed": None if info.kept is None else round(float(info.kept.float().mean()), 4)})
print(json.dumps({"config": cfg, "n": tr.num_gaussians(), "per_view": rows}))
