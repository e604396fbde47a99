#!/usr/bin/env python3
"""Golden vectors of the SDF term of FusionSense's training loss (``use_sdf_loss``), made by EXECUTING the reference's
own ``DNSplatterModel.get_loss_dict`` (/root/reference/dn_splatter/dn_model.py:838-882 with its helpers
``sample_points_in_gaussians`` :1445, ``get_ideal_sdf`` :1497, ``get_density`` :1575, ``get_sdf`` :1638,
``get_sdf_loss_weight`` :1668 and utils/camera_utils.py:project_pix) in this container, twice on the same inputs: with the
term off and with it on.  The stored value / gradients are the DIFFERENCE, i.e. exactly what ``sdf_loss_lambda * sdf_loss``
adds.  The term draws from torch's global generator inside the method: the two draws (``torch.multinomial`` picks,
``torch.randn`` offsets) are recorded as they happen and stored with the inputs, so that the restatement
(fusionsense_amd/sdf.py) can be run on the very same samples.

``quat_to_rotmat`` comes from gsplat 1.0.0 in the reference (absent here): the restated one of fusionsense_amd/legacy.py
(normalise, then the standard wxyz formula) is injected — the same stand-in on both sides of the comparison.

Runs only where /root/reference exists.  Nothing of the reference's source travels: only seeded inputs and the numbers
it produced (tests/golden/reference_sdf.npz)."""
import dataclasses
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_reference_goldens as base  # noqa: E402  (stub machinery)

OUT = os.path.join(HERE, "reference_sdf.npz")


def _splatfacto_standin():
    m = base._Stub("nerfstudio.models.splatfacto")
    m.__path__ = []

    @dataclasses.dataclass
    class SplatfactoModelConfig:
        pass

    class SplatfactoModel:
        def get_gt_img(self, image):
            return image.to(self.device)

        def get_loss_dict(self, outputs, batch, metrics_dict=None):
            z = torch.zeros((), device=self.device)
            return {"main_loss": z, "scale_reg": z.clone()}

        means = property(lambda self: self.gauss_params["means"])
        scales = property(lambda self: self.gauss_params["scales"])
        quats = property(lambda self: self.gauss_params["quats"])
        opacities = property(lambda self: self.gauss_params["opacities"])
        num_points = property(lambda self: self.gauss_params["means"].shape[0])

    m.SplatfactoModel = SplatfactoModel
    m.SplatfactoModelConfig = SplatfactoModelConfig
    return m


def case(model_mod, losses, seed, H, W, n_gauss, with_mask, with_anchors, k):
    g = torch.Generator().manual_seed(seed)
    sys.path.insert(0, os.path.join(base.REF, "configs"))
    import config as ref_cfg

    def make_cfg(use_sdf):
        return types.SimpleNamespace(
            use_depth_loss=ref_cfg.use_depth_loss, sensor_depth_lambda=ref_cfg.sensor_depth_lambda, depth_tolerance=0.1,
            depth_loss_type=losses.DepthLossType.EdgeAwareLogL1, mono_depth_lambda=0.2,
            use_depth_smooth_loss=ref_cfg.use_depth_smooth_loss, smooth_loss_type=losses.DepthLossType.TV,
            smooth_loss_lambda=0.1, use_normal_loss=ref_cfg.use_normal_loss, normal_supervision=ref_cfg.normal_supervision,
            use_normal_cosine_loss=False, use_normal_tv_loss=True, normal_lambda=ref_cfg.normal_lambda,
            two_d_gaussians=True, use_sparse_loss=False, sparse_lambda=0.1, sparse_loss_steps=10,
            use_sdf_loss=use_sdf, sdf_loss_lambda=0.1, apply_sdf_loss_after_iters=10, apply_sdf_loss_iters=5,
            num_sdf_samples=20, knn_to_track=k, reset_alpha_every=30, refine_every=100)

    # the scene: Gaussians in a slab in front of a camera at the origin that looks down -z (nerfstudio's convention)
    means0 = torch.stack([2.0 * torch.rand(n_gauss, generator=g) - 1.0, 1.2 * torch.rand(n_gauss, generator=g) - 0.6,
                          -(2.5 + 1.5 * torch.rand(n_gauss, generator=g))], dim=-1)
    scales0 = torch.randn(n_gauss, 3, generator=g) * 0.4 - 2.5
    quats0 = torch.randn(n_gauss, 4, generator=g)
    opac0 = torch.randn(n_gauss, 1, generator=g) * 1.5
    normals_world = torch.nn.functional.normalize(torch.randn(n_gauss, 3, generator=g), dim=-1)
    d2 = torch.cdist(means0, means0)
    knn = torch.argsort(d2, dim=-1)[:, 1:k + 1].contiguous()  # knn_sk: k + 1 nearest, the first (itself) dropped
    vis = torch.nonzero(torch.rand(n_gauss, generator=g) < 0.8).reshape(-1)
    add_mask = None
    if with_anchors:
        add_mask = torch.zeros(n_gauss, dtype=torch.bool)
        add_mask[-7:] = True
    fx, cx, cy = 30.0, W / 2.0, H / 2.0
    c2w = torch.eye(4)[None, :3].clone()
    c2w[0, :3, 3] = torch.tensor([0.05, -0.03, 0.1])
    rgb = torch.rand(H, W, 3, generator=g)
    depth0 = 3.0 + 0.6 * torch.rand(H, W, 1, generator=g)
    normal0 = torch.rand(H, W, 3, generator=g)
    image = torch.rand(H, W, 3, generator=g)
    sensor = 0.3 + 2.0 * torch.rand(H, W, 1, generator=g)
    normal_gt = torch.rand(H, W, 3, generator=g)
    mask = (torch.rand(H, W, 1, generator=g) > 0.25) if with_mask else None
    results, draws = {}, {}
    for use_sdf in (False, True):
        P = {n: t.clone().requires_grad_(True) for n, t in
             dict(means=means0, scales=scales0, quats=quats0, opacities=opac0).items()}
        self = object.__new__(model_mod.DNSplatterModel)
        self.config = make_cfg(use_sdf)
        self.device = torch.device("cpu")
        self.step = 15  # > apply_sdf_loss_after_iters, a multiple of apply_sdf_loss_iters, no multiple of 100
        self.depth_loss = losses.DepthLoss(self.config.depth_loss_type)
        self.smooth_loss = losses.DepthLoss(depth_loss_type=losses.DepthLossType.TV)
        self.tv_loss = losses.TVLoss()
        self.gauss_params = dict(P, normals=normals_world)
        self._knn = knn
        self.vis_indices = vis
        self.add_mask = add_mask
        if add_mask is not None:
            tn = torch.nn.functional.normalize(torch.randn(7, 3, generator=torch.Generator().manual_seed(seed + 100)), dim=-1)
            self.kwargs = {"metadata": {"touch_patches": [{"normals": tn[:3]}, {"normals": tn[3:]}]}}
        t1 = lambda v: torch.tensor([[float(v)]])  # noqa: E731
        self.camera = types.SimpleNamespace(camera_to_worlds=c2w, fx=t1(fx), fy=t1(fx * 1.1), cx=t1(cx), cy=t1(cy),
                                            width=torch.tensor([[W]]), height=torch.tensor([[H]]))
        depth = depth0.clone().requires_grad_(True)
        normal = normal0.clone().requires_grad_(True)
        batch = {"image": image.clone(), "sensor_depth": sensor.clone(), "normal": normal_gt.clone()}
        if mask is not None:
            batch["mask"] = mask
        outputs = {"rgb": rgb, "depth": depth, "normal": normal, "background": torch.ones(3)}
        real_multinomial, real_randn = torch.multinomial, torch.randn

        def rec_multinomial(*a, **kw):
            r = real_multinomial(*a, **kw)
            draws["picks"] = r.clone()
            return r

        def rec_randn(*a, **kw):
            r = real_randn(*a, **kw)
            draws["centered"] = r.clone()
            return r
        torch.manual_seed(seed)
        torch.multinomial, torch.randn = rec_multinomial, rec_randn
        try:
            ld = model_mod.DNSplatterModel.get_loss_dict(self, outputs, batch)
        finally:
            torch.multinomial, torch.randn = real_multinomial, real_randn
        loss = ld["main_loss"] + ld["scale_reg"]
        loss.backward()
        results[use_sdf] = dict(loss=loss.detach().double(), depth=depth.grad,
                                **{n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in P.items()})
    on, off = results[True], results[False]
    assert "picks" in draws and "centered" in draws, "the SDF term did not run"
    assert float((on["loss"] - off["loss"]).abs()) > 0
    assert on["depth"] is None or torch.equal(on["depth"], off["depth"]), "the SDF term must not reach the depth image"
    out = dict(means=means0, scales=scales0, quats=quats0, opacities=opac0, knn=knn, vis_indices=vis,
               has_anchors=np.array(with_anchors), add_mask=add_mask if add_mask is not None else torch.zeros(n_gauss, dtype=torch.bool),
               depth=depth0, c2w=c2w[0], intr=np.array([fx, cx, cy, W, H], dtype=np.float64),
               has_mask=np.array(with_mask), mask=mask if mask is not None else torch.ones(H, W, 1, dtype=torch.bool),
               picks=draws["picks"], centered=draws["centered"],
               cfg=np.array([20, 0.1], dtype=np.float64),  # num_sdf_samples, sdf_loss_lambda
               term=(on["loss"] - off["loss"]).float())
    for n in ("means", "scales", "quats", "opacities"):
        out["v_" + n] = on[n] - off[n]
    return out


def main():
    if not os.path.isdir(base.REF):
        raise SystemExit("reference not present; goldens can only be regenerated in the authoring container")
    base._install_stubs()
    sys.modules["nerfstudio.models.splatfacto"] = _splatfacto_standin()
    sys.modules["nerfstudio.models"].splatfacto = sys.modules["nerfstudio.models.splatfacto"]
    losses = base._load(os.path.join(base.REF, "dn_splatter", "losses.py"), "dn_splatter.losses")
    model = base._load(os.path.join(base.REF, "dn_splatter", "dn_model.py"), "dn_splatter.dn_model")
    cu = base._load(os.path.join(base.REF, "dn_splatter", "utils", "camera_utils.py"), "dn_splatter.utils.camera_utils")
    model.project_pix = cu.project_pix
    from fusionsense_amd.legacy import quat_to_rotmat  # (gsplat 1.0.0's, restated: the package is not in this container)
    model.quat_to_rotmat = quat_to_rotmat
    out = {}
    cases = {"plain": dict(seed=21, H=24, W=40, n_gauss=60, with_mask=False, with_anchors=False, k=4),
             "masked": dict(seed=22, H=30, W=36, n_gauss=80, with_mask=True, with_anchors=True, k=6)}
    for name, kw in cases.items():
        for key, v in case(model, losses, **kw).items():
            out[f"{name}.{key}"] = v.numpy() if torch.is_tensor(v) else v
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: float(v) for k, v in out.items() if k.endswith(".term")},
          {k: int(v.sum()) for k, v in out.items() if k.endswith(".picks")})


if __name__ == "__main__":
    main()
