#!/usr/bin/env python3
"""Generate golden vectors by EXECUTING the reference's own Python helpers in this container.

Runs only where /root/reference exists (the authoring container).  The reference cannot travel
to the GPU box, so the inputs/outputs are committed as tests/golden/reference_helpers.npz and
this script is committed next to them.  Nothing of the reference's source is copied: the
functions are imported from where they lie, with stub modules standing in for the heavy
third-party imports at the top of the files (nerfstudio, gsplat, torchmetrics, open3d ...),
none of which the executed helpers touch.

Helpers pinned (all in /root/reference/dn_splatter):
  dn_model.py:1996  points_in_non_aabb        dn_model.py:2035  random_quat_tensor
  dn_model.py:2050  SH2RGB                    dn_model.py:2058  rotate_vector_to_vector
  dn_model.py:2092  matrix_to_quaternion      dn_model.py:2153  invert_quaternion
  losses.py:177     EdgeAwareLogL1            losses.py:269     TVLoss
  losses.py:241     EdgeAwareTV               losses.py:161     LogL1
"""
import dataclasses
import importlib.util
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_helpers.npz")


class _Stub(types.ModuleType):
    """A module whose every attribute is a permissive dummy class / callable."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        @dataclasses.dataclass
        class Dummy:
            def __init__(self, *a, **k):
                pass

            def __call__(self, *a, **k):
                return self

        Dummy.__name__ = name
        setattr(self, name, Dummy)
        return Dummy


def _install_stubs():
    names = [
        "torchvision", "torchvision.transforms", "torchvision.transforms.functional",
        "torchmetrics", "torchmetrics.image", "torchmetrics.image.lpip",
        "gsplat", "gsplat.rendering", "gsplat.cuda_legacy", "gsplat.cuda_legacy._torch_impl",
        "gsplat.cuda_legacy._wrapper",
        "nerfstudio", "nerfstudio.cameras", "nerfstudio.cameras.camera_optimizers",
        "nerfstudio.cameras.cameras", "nerfstudio.data", "nerfstudio.data.scene_box",
        "nerfstudio.engine", "nerfstudio.engine.callbacks", "nerfstudio.engine.optimizers",
        "nerfstudio.models", "nerfstudio.models.splatfacto", "nerfstudio.utils",
        "nerfstudio.utils.colors", "nerfstudio.utils.rich_utils", "nerfstudio.field_components",
        "nerfstudio.field_components.field_heads",
        "dn_splatter", "dn_splatter.metrics", "dn_splatter.utils", "dn_splatter.utils.camera_utils",
        "dn_splatter.utils.knn", "dn_splatter.utils.normal_utils", "dn_splatter.utils.utils",
        "dn_splatter.utils.cluster_high_grad_gs", "open3d", "cv2", "sklearn", "sklearn.neighbors",
    ]
    for n in names:
        if n not in sys.modules:
            m = _Stub(n)
            m.__path__ = []
            sys.modules[n] = m
        if "." in n:
            parent, child = n.rsplit(".", 1)
            setattr(sys.modules[parent], child, sys.modules[n])


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not present; goldens can only be regenerated in the authoring container")
    _install_stubs()
    losses = _load(os.path.join(REF, "dn_splatter", "losses.py"), "dn_splatter.losses")
    model = _load(os.path.join(REF, "dn_splatter", "dn_model.py"), "dn_splatter.dn_model")

    out = {}
    g = torch.Generator().manual_seed(0)

    # random_quat_tensor: consumes torch's global RNG -> pin via manual_seed
    torch.manual_seed(123)
    out["rq_seed"] = np.array(123)
    out["rq_out"] = model.random_quat_tensor(257).numpy()

    sh = torch.randn(50, 3, generator=g)
    out["sh2rgb_in"] = sh.numpy()
    out["sh2rgb_out"] = model.SH2RGB(sh).numpy()

    # points_in_non_aabb: a rotated box, vertices in the reference's corner convention
    # (0 -> 1, 0 -> 2, 0 -> 4 are the three edge directions)
    ax = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    ext = torch.tensor([0.3, 0.2, 0.1])
    c = torch.tensor([0.1, -0.2, 0.05])
    corners = []
    for i in range(8):
        s = torch.tensor([(i >> 0) & 1, (i >> 1) & 1, (i >> 2) & 1], dtype=torch.float32) * 2 - 1
        corners.append(c + (ax * (s * ext)[None, :]).sum(-1))
    box = torch.stack(corners)
    pts = torch.rand(500, 3, generator=g) - 0.5
    out["aabb_box"] = box.numpy()
    out["aabb_pts"] = pts.numpy()
    out["aabb_mask"] = model.points_in_non_aabb(pts, box).numpy()

    v1 = torch.randn(40, 3, generator=g)
    v2 = torch.randn(40, 3, generator=g)
    v2[0] = v1[0] * 2.0          # same direction
    v2[1] = -v1[1] * 0.5         # opposite direction
    out["rvv_v1"], out["rvv_v2"] = v1.numpy(), v2.numpy()
    R = model.rotate_vector_to_vector(v1, v2)
    out["rvv_R"] = R.numpy()
    # matrix_to_quaternion on proper rotations hitting all four branches
    from math import cos, sin, pi
    mats = [torch.eye(3)]
    for axis, ang in ((0, pi * 0.9), (1, pi * 0.95), (2, pi * 0.99), (0, 0.3), (1, 1.2), (2, 2.0)):
        ca, sa = cos(ang), sin(ang)
        m = torch.eye(3)
        i, j = [(1, 2), (2, 0), (0, 1)][axis]
        m[i, i], m[i, j], m[j, i], m[j, j] = ca, -sa, sa, ca
        mats.append(m)
    mats = torch.stack(mats + [R[5], R[9], R[17]])
    out["m2q_in"] = mats.numpy()
    out["m2q_out"] = model.matrix_to_quaternion(mats).numpy()
    q = torch.randn(11, 4, generator=g)
    out["invq_in"] = q.numpy()
    out["invq_out"] = model.invert_quaternion(q).numpy()

    # losses
    H, W = 24, 32
    pred = torch.rand(H, W, 1, generator=g) * 3
    gt = torch.rand(H, W, 1, generator=g) * 3
    rgb = torch.rand(H, W, 3, generator=g)
    mask = torch.rand(H, W, 1, generator=g) > 0.3
    out["loss_pred"], out["loss_gt"], out["loss_rgb"], out["loss_mask"] = (
        pred.numpy(), gt.numpy(), rgb.numpy(), mask.numpy())
    ea = losses.EdgeAwareLogL1()
    out["edge_logl1_nomask"] = ea(pred, gt, rgb, None).numpy()
    out["edge_logl1_mask"] = ea(pred, gt, rgb, mask).numpy()
    out["logl1"] = losses.LogL1()(pred, gt).numpy()
    out["tv_depth"] = losses.TVLoss()(pred).numpy()
    out["tv_rgb"] = losses.TVLoss()(rgb).numpy()
    out["edge_tv"] = losses.EdgeAwareTV()(pred, rgb).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
