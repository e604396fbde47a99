#!/usr/bin/env python3
"""Golden vectors made by EXECUTING the reference's ``DNSplatterModel.compute_level_surface_points``
(/root/reference/dn_splatter/dn_model.py:1706-1946: the level-set extraction behind ``gs-mesh``'s LevelSetExtractor,
export_mesh.py:486-600) in this container, the way make_reference_model_goldens.py executes get_outputs: the method is
called unbound on a bare instance, its own helpers (scale_rot_to_inv_cov3d, invert_quaternion, and
utils/camera_utils.get_colored_points_from_depth) are the reference's, the rasterizer under its ``get_outputs`` call is
this repo's CPU oracle, ``knn_sk`` is an exact brute-force stand-in with sklearn's drop-the-first-column semantics, and
``random.sample`` is pinned to the identity so that the points come out in pixel order.

Runs only where /root/reference exists; writes tests/golden/reference_levelset.npz (inputs and outputs only).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_reference_goldens as base  # noqa: E402
import make_reference_model_goldens as mg  # noqa: E402

OUT = os.path.join(HERE, "reference_levelset.npz")


class LevelCameras(mg.FakeCameras):
    """+ what compute_level_surface_points reads: fx / fy / cx / cy as one-element tensors."""

    def __init__(self, c2w, fx, fy, cx, cy, W, H):
        super().__init__(c2w, fx, fy, cx, cy, W, H)
        self.fx, self.fy = torch.tensor([[fx]]), torch.tensor([[fy]])
        self.cx, self.cy = torch.tensor([[cx]]), torch.tensor([[cy]])


def main():
    mm = mg.load_reference_model()
    cu = base._load(os.path.join(base.REF, "dn_splatter", "utils", "camera_utils.py"), "dn_splatter.utils.camera_utils")
    mm.get_colored_points_from_depth = cu.get_colored_points_from_depth

    def knn_drop_first(x, y, k):  # utils/knn.py:29-44: k + 1 neighbours, the first column dropped
        return torch.cdist(y.double(), x.double()).topk(k + 1, largest=False).indices[:, 1:]
    mm.knn_sk = knn_drop_first
    mm.random = types.SimpleNamespace(sample=lambda pop, k: list(pop)[:k])  # identity: pixel order
    from fusionsense_amd import scenes

    out = {}
    # a cube of 900 disc-like Gaussians, opaque enough for the density to cross the three levels
    params, cam = scenes.cube_scene(900, seed=31)
    g = torch.Generator().manual_seed(4)
    params["scales"] = torch.log(0.02 + 0.05 * torch.rand(900, 3, generator=g))
    params["scales"][:, 2] += np.log(0.15)
    params["opacities"] = 1.5 + torch.randn(900, 1, generator=g)
    W, H = 48, 36
    intr = (52.0, 50.0, 23.5, 18.5)
    fcam = LevelCameras(cam.c2w, *intr, W, H)
    for k, v in params.items():
        out[f"in.{k}"] = v.numpy()
    out["c2w"], out["intr"] = cam.c2w.numpy(), np.array(intr + (W, H))
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    mask = (((xx - 24) / 20.0) ** 2 + ((yy - 18) / 15.0) ** 2 < 1.0)[..., None]
    out["mask"] = mask.numpy()
    for case, mode, use_mask, n_samples in (("closest", "closest_gaussian", False, 10 ** 9), ("analytical", "analytical", True, 10 ** 9),
                                           ("few", "closest_gaussian", False, 40)):
        m = mg.make_model(mm, params, 100, dict(knn_to_track=16, use_binary_opacities=False))
        m.training = False
        type(m).normals = property(lambda s: s.gauss_params["normals"])
        type(m).device = torch.device("cpu")
        res = m.compute_level_surface_points(fcam, n_samples, mask if use_mask else None, (0.1, 0.3, 0.5), mode)
        if case == "closest":  # the render the extraction started from (oracle rasterizer) + the Gaussians' normals
            o = m.get_outputs(fcam)
            out["render.depth"], out["render.rgb"] = o["depth"].detach().numpy(), o["rgb"].detach().numpy()
            out["render.normals_world"] = m.gauss_params["normals"].detach().numpy()
        for lv, d in res.items():
            for key in ("points", "normals", "colors"):
                out[f"{case}.{lv}.{key}"] = d[key].detach().numpy()
        out[f"{case}.mode"] = np.array(["closest_gaussian", "analytical"].index(mode))
        out[f"{case}.use_mask"], out[f"{case}.num_samples"] = np.array(use_mask), np.array(min(n_samples, 2 ** 31 - 1))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays;", {k: out[k].shape for k in out if k.endswith(".points")})


if __name__ == "__main__":
    main()
