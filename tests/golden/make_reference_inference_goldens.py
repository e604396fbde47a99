#!/usr/bin/env python3
"""Golden vectors for the inference callers, made by EXECUTING the reference's own helpers:
  dn_splatter/metrics.py        DepthMetrics, NormalMetrics, mean_angular_error   (the eval loop's metrics,
                                dn_model.py:1032-1148 / dn_pipeline.py:233-253)
  dn_splatter/utils/camera_utils.py   get_camera_coords, get_means3d_backproj, project_pix,
                                get_colored_points_from_depth                     (mesh / point-cloud back-projection,
                                export_mesh.py:355-400, 543, 836)
  dn_splatter/export_mesh.py    find_depth_edges                                  (depth-edge filter, :59-91)
Runs only where /root/reference exists; only inputs and outputs are stored (tests/golden/reference_inference.npz)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_reference_goldens as base  # noqa: E402

OUT = os.path.join(HERE, "reference_inference.npz")


def main():
    if not os.path.isdir(base.REF):
        raise SystemExit("reference not present")
    base._install_stubs()
    import types

    class _Sub:  # tyro.conf.FlagConversionOff[...] at the bottom of export_mesh.py
        def __class_getitem__(cls, item):
            return item

    tyro = types.ModuleType("tyro")
    tyro.conf = types.SimpleNamespace(FlagConversionOff=_Sub, subcommand=lambda **k: None)
    tyro.extras = types.SimpleNamespace(set_accent_color=lambda *a: None)
    tyro.cli = lambda *a, **k: None
    sys.modules["tyro"] = tyro
    for n in ("tqdm", "nerfstudio.utils.eval_utils"):
        if n not in sys.modules:
            m = base._Stub(n); m.__path__ = []; sys.modules[n] = m
    cam = base._load(os.path.join(base.REF, "dn_splatter", "utils", "camera_utils.py"), "dn_splatter.utils.camera_utils")
    sys.modules["dn_splatter.utils"].camera_utils = cam
    metrics = base._load(os.path.join(base.REF, "dn_splatter", "metrics.py"), "dn_splatter.metrics")
    mesh = base._load(os.path.join(base.REF, "dn_splatter", "export_mesh.py"), "dn_splatter.export_mesh")
    g = torch.Generator().manual_seed(0)
    out = {}
    H, W = 14, 18
    pred = 0.2 + 2.0 * torch.rand(1, H, W, generator=g)
    gt = 0.2 + 2.0 * torch.rand(1, H, W, generator=g)
    gt[torch.rand(1, H, W, generator=g) < 0.2] = 0.05  # below the tolerance: excluded
    out["dm_pred"], out["dm_gt"] = pred.numpy(), gt.numpy()
    out["dm_out"] = np.array([float(x) for x in metrics.DepthMetrics()(pred, gt)])
    pn = torch.nn.functional.normalize(torch.randn(1, 3, H, W, generator=g), dim=1)
    gn = torch.nn.functional.normalize(torch.randn(1, 3, H, W, generator=g), dim=1)
    out["nm_pred"], out["nm_gt"] = pn.numpy(), gn.numpy()
    out["nm_out"] = np.array([float(x) for x in metrics.NormalMetrics()(pn, gn)])
    out["mae_map"] = metrics.mean_angular_error(pn, gn).numpy()
    # back-projection
    depth = 0.5 + torch.rand(H, W, 1, generator=g)
    rgb = torch.rand(H, W, 3, generator=g)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    c2w = torch.eye(4)
    c2w[:3, :3] = q
    c2w[:3, 3] = torch.randn(3, generator=g)
    c2w34 = c2w[:3, :4]
    fx, fy, cx, cy = 20.0, 21.0, 9.3, 6.6
    out["bp_depth"], out["bp_rgb"], out["bp_c2w"] = depth.numpy(), rgb.numpy(), c2w34.numpy()
    out["bp_intr"] = np.array([fx, fy, cx, cy])
    out["bp_coords"] = cam.get_camera_coords((W, H)).numpy()
    pts, coords = cam.get_means3d_backproj(depths=depth, fx=fx, fy=fy, cx=cx, cy=cy, img_size=(W, H), c2w=c2w34,
                                           device=torch.device("cpu"))
    out["bp_points"] = pts.numpy()
    idx = torch.randperm(H * W, generator=g)[:40]
    out["bp_idx"] = idx.numpy()
    p2, c2 = cam.get_colored_points_from_depth(depths=depth, rgbs=rgb, c2w=c2w34, fx=fx, fy=fy, cx=cx, cy=cy,
                                               img_size=(W, H), mask=idx)
    out["bp_points_sel"], out["bp_colors_sel"] = p2.numpy(), c2.numpy()
    uvz = cam.project_pix(pts, fx, fy, cx, cy, c2w34, torch.device("cpu"), return_z_depths=True)
    out["bp_uvz"] = uvz.numpy()
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    d2 = (1.0 + 0.01 * xx + 0.02 * yy).float()[..., None]  # a smooth ramp ...
    d2[4:9, 5:12] += 1.5  # ... with a depth step
    out["edge_depth"] = d2.numpy()
    out["edge_out"] = mesh.find_depth_edges(d2, threshold=0.01, dilation_itr=3).numpy()
    out["edge_out_1"] = mesh.find_depth_edges(d2, threshold=0.05, dilation_itr=1).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
