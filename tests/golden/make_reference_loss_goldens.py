#!/usr/bin/env python3
"""Golden vectors of the FusionSense training loss, made by EXECUTING the reference's own
``DNSplatterModel.get_loss_dict`` (/root/reference/dn_splatter/dn_model.py:673-925) in this container.

The method is called unbound on a bare instance whose base class is a small stand-in for nerfstudio's
``SplatfactoModel`` (nerfstudio is not installed): the stand-in's ``get_loss_dict`` returns a ZERO main loss, so
the stored value is exactly what FusionSense's own code adds on top of splatfacto's photometric term — mask
handling, EdgeAwareLogL1 on the sensor depth with the valid mask, TV on depth, normal L1 + TV, the min-scale term
and the touch-normal MSE — with the weights of /root/reference/configs/config.py.  Gradients with respect to the
rendered depth / normal images and the log-scales come from autograd through that same code.

Runs only where /root/reference exists.  Nothing of the reference's source travels: only the seeded inputs and the
numbers it produced (tests/golden/reference_loss.npz), next to this script.
"""
import dataclasses
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_reference_goldens as base  # noqa: E402  (stub machinery)

OUT = os.path.join(HERE, "reference_loss.npz")


def _splatfacto_standin():
    """nerfstudio.models.splatfacto with a SplatfactoModel that supplies what get_loss_dict touches."""
    m = base._Stub("nerfstudio.models.splatfacto")
    m.__path__ = []

    @dataclasses.dataclass
    class SplatfactoModelConfig:
        pass

    class SplatfactoModel:
        def get_gt_img(self, image):
            return image.to(self.device)  # float images, num_downscales = 0 (dn_model.py:118)

        def get_loss_dict(self, outputs, batch, metrics_dict=None):
            z = torch.zeros((), device=self.device)
            return {"main_loss": z, "scale_reg": z.clone()}

        @property
        def scales(self):
            return self.gauss_params["scales"]

        @property
        def opacities(self):
            return self.gauss_params["opacities"]

    m.SplatfactoModel = SplatfactoModel
    m.SplatfactoModelConfig = SplatfactoModelConfig
    return m


def case(model_mod, losses, seed, H, W, n_gauss, with_mask, with_touch, holes, mono=None, normal_supervision=None,
         smooth="TV", cosine=False, depth_type="EdgeAwareLogL1", intr=None, sparse=None):
    """Round 4: the switches that are OFF in FusionSense's configuration but reachable from scripts/train.py:117-145 /
    the model config — ``normal_supervision="depth"`` (pseudo normals from the rendered depth, dn_model.py:774-795 with
    utils/normal_utils.py:8-46 executed as is), ``smooth="EdgeAwareTV"`` (:757-760, losses.py:241-266), ``cosine``
    (:807-813 with metrics.mean_angular_error) and the plain depth-loss types L1 / LogL1 / MSE / HuberL1 (:731-736).
    ``sparse`` = the model's step: the sparse-opacity term (:821-837; entropy of the visible Gaussians' opacities, every
    ``sparse_loss_steps``-th step outside the 100 steps after an opacity reset) is on, the opacity logits and the
    visible indices are part of the case."""
    g = torch.Generator().manual_seed(seed)
    cfgd = {}
    sys.path.insert(0, os.path.join(base.REF, "configs"))
    import config as ref_cfg  # /root/reference/configs/config.py: the values scripts/train.py passes on
    cfg = types.SimpleNamespace(
        use_depth_loss=ref_cfg.use_depth_loss, sensor_depth_lambda=ref_cfg.sensor_depth_lambda,
        depth_tolerance=0.1, depth_loss_type=getattr(losses.DepthLossType, depth_type), mono_depth_lambda=0.2,
        use_depth_smooth_loss=ref_cfg.use_depth_smooth_loss, smooth_loss_type=getattr(losses.DepthLossType, smooth),
        smooth_loss_lambda=0.1, use_normal_loss=ref_cfg.use_normal_loss,
        normal_supervision=normal_supervision or ref_cfg.normal_supervision,
        use_normal_cosine_loss=bool(cosine), use_normal_tv_loss=True, normal_lambda=ref_cfg.normal_lambda,
        two_d_gaussians=True, use_sparse_loss=sparse is not None, sparse_lambda=0.1, sparse_loss_steps=10,
        use_sdf_loss=False, sdf_loss_lambda=0.1, reset_alpha_every=30, refine_every=100)
    self = object.__new__(model_mod.DNSplatterModel)
    self.config = cfg
    self.device = torch.device("cpu")
    self.step = 7 if sparse is None else int(sparse)  # (not a multiple of 100: the reference writes a debug JPEG there)
    self.depth_loss = losses.DepthLoss(cfg.depth_loss_type)
    self.smooth_loss = losses.DepthLoss(depth_loss_type=getattr(losses.DepthLossType, smooth))  # (dn_model.py:239-242)
    if intr is not None:  # what the "depth" normal supervision reads from the last get_outputs' camera (:775-786)
        t1 = lambda v: torch.tensor([[float(v)]])  # noqa: E731
        self.camera = types.SimpleNamespace(camera_to_worlds=torch.eye(4)[None, :3], fx=t1(intr[0]), fy=t1(intr[1]),
                                            cx=t1(intr[2]), cy=t1(intr[3]), width=torch.tensor([[W]]),
                                            height=torch.tensor([[H]]))
    self.tv_loss = losses.TVLoss()
    scales = (torch.randn(n_gauss, 3, generator=g) * 0.7 - 4.0).requires_grad_(True)
    normals_world = torch.nn.functional.normalize(torch.randn(n_gauss, 3, generator=g), dim=-1)
    self.gauss_params = {"scales": scales, "normals": normals_world}
    opac = vis = None
    if sparse is not None:
        opac = (torch.randn(n_gauss, 1, generator=g) * 2.0).requires_grad_(True)
        vis = torch.nonzero(torch.rand(n_gauss, generator=g) < 0.7).reshape(-1)
        self.gauss_params["opacities"] = opac
        self.vis_indices = vis
    n_touch = 9 if with_touch else 0
    if with_touch:
        add_mask = torch.zeros(n_gauss, dtype=torch.bool)
        add_mask[-n_touch:] = True
        self.add_mask = add_mask
        tn = torch.nn.functional.normalize(torch.randn(n_touch, 3, generator=g), dim=-1)
        self.kwargs = {"metadata": {"touch_patches": [{"normals": tn[:4]}, {"normals": tn[4:]}]}}
    else:
        self.add_mask = None
        tn = torch.zeros(0, 3)
    rgb = torch.rand(H, W, 3, generator=g)
    depth = (0.3 + 2.0 * torch.rand(H, W, 1, generator=g)).requires_grad_(True)
    normal = torch.rand(H, W, 3, generator=g).requires_grad_(True)
    image = torch.rand(H, W, 3, generator=g)
    image[: H // 4] *= 0.02  # dark rows: the clamp(min=10/255) of dn_model.py:692 is active
    sensor = 0.3 + 2.0 * torch.rand(H, W, 1, generator=g)
    if holes:
        sensor[torch.rand(H, W, 1, generator=g) < 0.25] = 0.0  # invalid sensor depth (<= depth_tolerance)
    normal_gt = torch.rand(H, W, 3, generator=g)
    batch = {"image": image.clone(), "sensor_depth": sensor.clone(), "normal": normal_gt.clone()}
    mono_depth = None
    if mono is not None:
        # the monocular-depth branch (dn_model.py:737-750; off in FusionSense, whose batches carry no "mono_depth"):
        # "both" = next to the sensor depth, "only" = instead of it
        mono_depth = 0.2 + 2.5 * torch.rand(H, W, 1, generator=g)
        mono_depth[torch.rand(H, W, 1, generator=g) < 0.2] = 0.05  # (<= depth_tolerance: invalid)
        batch["mono_depth"] = mono_depth.clone()
        if mono == "only":
            del batch["sensor_depth"]
    if with_mask:
        mask = torch.rand(H, W, 1, generator=g) > 0.3
        batch["mask"] = mask
    else:
        mask = torch.ones(H, W, 1, dtype=torch.bool)
    outputs = {"rgb": rgb, "depth": depth, "normal": normal, "background": torch.ones(3)}
    ld = model_mod.DNSplatterModel.get_loss_dict(self, outputs, batch)
    loss = ld["main_loss"] + ld["scale_reg"]
    loss.backward()
    return dict(rgb=rgb, depth=depth.detach(), normal=normal.detach(), image=image, sensor_depth=sensor,
                normal_gt=normal_gt, mask=mask, has_mask=np.array(with_mask), scales=scales.detach(),
                normals_world=normals_world, n_touch=np.array(n_touch), touch_normals=tn,
                loss=loss.detach(), v_depth=depth.grad, v_normal=normal.grad, v_scales=scales.grad,
                **({} if mono is None else dict(mono_depth=mono_depth, mono_only=np.array(mono == "only"),
                                                mono_depth_lambda=np.array(cfg.mono_depth_lambda, dtype=np.float32))),
                **({} if intr is None else dict(intr=np.array(intr, dtype=np.float64))),
                **({} if sparse is None else dict(
                    opacities=opac.detach(), vis_indices=vis, v_opacities=opac.grad if opac.grad is not None else torch.zeros_like(opac),
                    sparse=np.array([int(sparse), cfg.sparse_loss_steps, cfg.reset_alpha_every, cfg.refine_every], dtype=np.int32),
                    sparse_lambda=np.array(cfg.sparse_lambda, dtype=np.float32))),
                **({} if (normal_supervision is None and smooth == "TV" and not cosine and depth_type == "EdgeAwareLogL1")
                   else dict(switches=np.array([["mono", "depth"].index(normal_supervision or "mono"),
                                                ["TV", "EdgeAwareTV"].index(smooth), int(bool(cosine)),
                                                ["EdgeAwareLogL1", "L1", "LogL1", "MSE", "HuberL1"].index(depth_type)], dtype=np.int32))))  # (numeric codes)


def main():
    if not os.path.isdir(base.REF):
        raise SystemExit("reference not present; goldens can only be regenerated in the authoring container")
    base._install_stubs()
    sys.modules["nerfstudio.models.splatfacto"] = _splatfacto_standin()
    sys.modules["nerfstudio.models"].splatfacto = sys.modules["nerfstudio.models.splatfacto"]
    losses = base._load(os.path.join(base.REF, "dn_splatter", "losses.py"), "dn_splatter.losses")
    model = base._load(os.path.join(base.REF, "dn_splatter", "dn_model.py"), "dn_splatter.dn_model")
    # the real helpers of the "depth" normal supervision and of the cosine term (numpy + torch only)
    cu = base._load(os.path.join(base.REF, "dn_splatter", "utils", "camera_utils.py"), "dn_splatter.utils.camera_utils")
    nu = base._load(os.path.join(base.REF, "dn_splatter", "utils", "normal_utils.py"), "dn_splatter.utils.normal_utils")
    model.normal_from_depth_image = nu.normal_from_depth_image
    mt = sys.modules.get("dn_splatter.metrics")
    if mt is None or isinstance(mt, base._Stub):
        mt = types.ModuleType("dn_splatter.metrics")
        sys.modules["dn_splatter.metrics"] = mt
    import re
    src = open(os.path.join(base.REF, "dn_splatter", "metrics.py")).read()
    fn = re.search(r"^def mean_angular_error\(.*?(?=^\S)", src, re.S | re.M).group(0)  # (the function alone: the module imports torchmetrics)
    exec(compile("import torch\n" + fn, "dn_splatter/metrics.py:mean_angular_error", "exec"), mt.__dict__)
    out = {}
    cases = {"full": dict(seed=1, H=24, W=40, n_gauss=50, with_mask=True, with_touch=True, holes=True),
             "nomask": dict(seed=2, H=17, W=33, n_gauss=30, with_mask=False, with_touch=False, holes=True),
             "dense": dict(seed=3, H=32, W=32, n_gauss=64, with_mask=True, with_touch=True, holes=False),
             "mono": dict(seed=4, H=21, W=37, n_gauss=40, with_mask=True, with_touch=False, holes=True, mono="both"),
             "monoonly": dict(seed=5, H=19, W=26, n_gauss=20, with_mask=False, with_touch=False, holes=True, mono="only"),
             # ---- round 4: the remaining switches ----
             "ndepth": dict(seed=6, H=22, W=35, n_gauss=30, with_mask=True, with_touch=False, holes=True,
                            normal_supervision="depth", intr=(30.0, 28.0, 17.5, 11.0)),
             "ndepth_nomask": dict(seed=7, H=18, W=27, n_gauss=20, with_mask=False, with_touch=True, holes=False,
                                   normal_supervision="depth", intr=(25.0, 25.0, 13.0, 9.5)),
             "eatv": dict(seed=8, H=20, W=31, n_gauss=25, with_mask=True, with_touch=False, holes=True, smooth="EdgeAwareTV"),
             # (few rows: the row sums of the cosine term as called stay inside acos' domain — at real image heights the
             # clamp saturates and the term is constant; 12 rows = the smallest image the 11x11 SSIM window accepts)
             "cosine": dict(seed=9, H=12, W=29, n_gauss=20, with_mask=True, with_touch=False, holes=True, cosine=True),
             "cosine_depth": dict(seed=10, H=12, W=24, n_gauss=20, with_mask=False, with_touch=False, holes=False, cosine=True,
                                  normal_supervision="depth", intr=(20.0, 20.0, 12.0, 6.0)),
             "l1": dict(seed=11, H=19, W=30, n_gauss=20, with_mask=True, with_touch=False, holes=True, depth_type="L1"),
             "logl1": dict(seed=12, H=19, W=30, n_gauss=20, with_mask=False, with_touch=False, holes=True, depth_type="LogL1",
                           mono="both"),
             "mse": dict(seed=13, H=16, W=28, n_gauss=20, with_mask=True, with_touch=False, holes=True, depth_type="MSE",
                         smooth="EdgeAwareTV"),
             # ---- terms that stay torch code on top of the HIP node (integration.get_loss_dict_fused) ----
             "huber": dict(seed=14, H=18, W=29, n_gauss=20, with_mask=True, with_touch=False, holes=True,
                           depth_type="HuberL1", mono="both"),
             "sparse": dict(seed=15, H=17, W=26, n_gauss=40, with_mask=True, with_touch=True, holes=True, sparse=110),
             "sparse_off": dict(seed=16, H=17, W=26, n_gauss=40, with_mask=False, with_touch=False, holes=True, sparse=3050)}
    for name, kw in cases.items():
        for k, v in case(model, losses, **kw).items():
            out[f"{name}.{k}"] = v.numpy() if torch.is_tensor(v) else v
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: (v.shape, float(v)) if v.ndim == 0 else v.shape for k, v in out.items() if ".loss" in k or "v_" in k})


if __name__ == "__main__":
    main()
