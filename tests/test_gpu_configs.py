"""BASELINE.json configs #3 (FusionSense scene: RGB + sensor depth + mono normals + mask + touch patches, 1280x720),
#4 (6 M Gaussians, 1080p) and #5 (10 M Gaussians, half-precision attributes) on the HIP path.

Config #3's loss is the reference's own ``DNSplatterModel.get_loss_dict`` (dn_splatter/dn_model.py:673-925): the
kernels are checked against goldens made by EXECUTING that method (tests/golden/reference_loss.npz), against the
CPU oracle (oracle/loss_ref.py, itself pinned by the same goldens) on seeded inputs, end to end through the trainer
on a reduced scene, and at full size through size-independent properties.  All through the C-ABI."""
import math
import os

import numpy as np
import pytest
import torch

from fusionsense_amd import scenes
from helpers import rel_err

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "reference_loss.npz")


def _fb(batch, dev, cfg=None):
    from fusionsense_amd.losses import LossConfig, prepare_batch
    return prepare_batch(batch, cfg or LossConfig(), dev)


@pytest.mark.parametrize("case", ["full", "nomask", "dense"])
def test_fusion_loss_kernels_match_reference_goldens(dev, case):
    """ops._FusionLoss against numbers the reference's get_loss_dict produced itself: the gradient images of depth
    and normals, the log-scale gradient, and the value of everything FusionSense adds to the photometric term."""
    from fusionsense_amd import ops
    from fusionsense_amd.losses import LossConfig
    from oracle import loss_ref
    d = np.load(GOLD)
    t = lambda k: torch.from_numpy(d[f"{case}.{k}"])  # noqa: E731
    cfg = LossConfig()
    batch = {"image": t("image"), "sensor_depth": t("sensor_depth"), "normal": t("normal_gt")}
    if bool(d[f"{case}.has_mask"]):
        batch["mask"] = t("mask")
    fb = _fb(batch, dev, cfg)
    n_touch = int(d[f"{case}.n_touch"])
    N = t("scales").shape[0]
    touch_idx = torch.arange(N - n_touch, N, device=dev) if n_touch else None
    touch_normals = t("touch_normals").to(dev) if n_touch else None
    rgb = t("rgb").to(dev).requires_grad_(True)
    depth = t("depth").to(dev).requires_grad_(True)
    normal = t("normal").to(dev).requires_grad_(True)
    scales = t("scales").to(dev).requires_grad_(True)
    out = {"rgb": rgb, "depth": depth, "normal": normal, "normals_world": t("normals_world").to(dev)}
    loss = ops.fusion_loss(out, fb, cfg, scales, touch_idx, touch_normals)
    loss.backward()
    # the photometric share of the value comes from the oracle (it does not touch depth / normal / scales)
    photo = loss_ref.photometric_loss(t("rgb").double(), t("image").double(),
                                      t("mask").double() if "mask" in batch else None, cfg.ssim_lambda)
    assert abs(float(loss) - float(photo) - float(d[f"{case}.loss"])) < 2e-6 * max(1.0, abs(float(d[f"{case}.loss"])))
    for name, g in (("v_depth", depth.grad), ("v_normal", normal.grad), ("v_scales", scales.grad)):
        ref = t(name)
        assert torch.allclose(g.cpu(), ref, rtol=2e-5, atol=1e-9), (name, float((g.cpu() - ref).abs().max()))


@pytest.mark.parametrize("H,W,with_mask,with_touch", [(64, 96, True, True), (45, 77, True, False), (33, 32, False, True),
                                                       (720, 1280, True, True)])
def test_fusion_loss_matches_oracle(dev, H, W, with_mask, with_touch):
    """The whole get_loss_dict (photometric term included) against oracle/loss_ref in fp64 on seeded images: value
    and all four gradients; image sizes that are not multiples of the kernels' tiles, and config #3's own."""
    from fusionsense_amd import ops
    from fusionsense_amd.losses import LossConfig
    from oracle import loss_ref
    g = torch.Generator().manual_seed(H * 1000 + W)
    N, n_touch = 500, (37 if with_touch else 0)
    rgb = torch.rand(H, W, 3, generator=g)
    depth = 0.3 + 2.0 * torch.rand(H, W, 1, generator=g)
    normal = torch.rand(H, W, 3, generator=g)
    image = torch.rand(H, W, 3, generator=g)
    image[: H // 5] *= 0.03
    sensor = 0.3 + 2.0 * torch.rand(H, W, 1, generator=g)
    sensor[torch.rand(H, W, 1, generator=g) < 0.2] = 0.0
    normal_gt = torch.rand(H, W, 3, generator=g)
    scales = torch.randn(N, 3, generator=g) * 0.7 - 4.0
    nw = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    tn = torch.nn.functional.normalize(torch.randn(n_touch, 3, generator=g), dim=-1)
    batch = {"image": image, "sensor_depth": sensor, "normal": normal_gt}
    if with_mask:
        batch["mask"] = torch.rand(H, W, 1, generator=g) > 0.3
    add_mask = None
    if n_touch:
        add_mask = torch.zeros(N, dtype=torch.bool)
        add_mask[100:100 + n_touch] = True
    cfg = LossConfig()
    # oracle, fp64
    o = {k: v.double().requires_grad_(True) for k, v in (("rgb", rgb), ("depth", depth), ("normal", normal))}
    s64 = scales.double().requires_grad_(True)
    b64 = {k: (v if v.dtype == torch.bool else v.double()) for k, v in batch.items()}
    ld = loss_ref.get_loss_dict(o, b64, s64, nw.double(), add_mask, tn.double() if n_touch else None, loss_ref.LossConfig())
    ref = ld["main_loss"] + ld["scale_reg"]
    ref.backward()
    # HIP
    fb = _fb(batch, dev, cfg)
    h = {k: v.to(dev).requires_grad_(True) for k, v in (("rgb", rgb), ("depth", depth), ("normal", normal))}
    hs = scales.to(dev).requires_grad_(True)
    h["normals_world"] = nw.to(dev)
    ti = torch.nonzero(add_mask).squeeze(-1).to(dev) if n_touch else None
    loss = ops.fusion_loss(h, fb, cfg, hs, ti, tn.to(dev) if n_touch else None)
    loss.backward()
    assert abs(float(loss) - float(ref)) < 5e-6 * abs(float(ref)), (float(loss), float(ref))
    # fp32 tolerance of the loss gradients: 1e-4 of each tensor's own maximum (sign functions at |x| ~ 1e-7 aside)
    for k in ("rgb", "depth", "normal"):
        err = (h[k].grad.cpu().double() - o[k].grad).abs()
        scale = float(o[k].grad.abs().max())
        assert float((err > 1e-4 * scale).double().mean()) < 1e-5, (k, float(err.max()), scale)
    assert rel_err(hs.grad, s64.grad) < 1e-5
    # a second call with the seed known up front (the trainer's tape-free order) gives the same gradient images
    ctx_like = __import__("fusionsense_amd.fused", fromlist=["_DirectCtx"])._DirectCtx((True, True, True, False) + (False,) * 7)
    seed = torch.ones((), device=dev)
    with torch.no_grad():
        l2 = ops._FusionLoss.forward(ctx_like, h["rgb"].detach(), h["depth"].detach(), h["normal"].detach(), hs.detach(), fb,
                                     cfg, h["normals_world"], ti, tn.to(dev) if n_touch else None, seed, True)
        v = ops._FusionLoss.backward(ctx_like, seed)
    assert abs(float(l2) - float(loss)) < 1e-6 * abs(float(loss))
    assert torch.equal(v[0], h["rgb"].grad) and torch.equal(v[1], h["depth"].grad) and torch.equal(v[2], h["normal"].grad)


def _small_fusionsense(dev, W=160, H=96, n_hull=1200, n_bg=2400, seed=0):
    """Config #3 at reduced size: the same scene recipe, camera intrinsics scaled to a small frame."""
    params, cams = scenes.fusionsense_like_scene(seed=seed, n_hull=n_hull, n_bg=n_bg)
    g = torch.Generator().manual_seed(seed + 7)
    params["features_rest"] = 0.05 * torch.randn(params["features_rest"].shape, generator=g)
    params["opacities"] = params["opacities"] + 2.0  # visible from the first step
    sx, sy = W / 1280.0, H / 720.0
    cams = [scenes.Camera(c.c2w, c.fx * sx, c.fy * sy, c.cx * sx, c.cy * sy, W, H) for c in cams]
    return params, cams


def _synthetic_batch(H, W, seed):
    g = torch.Generator().manual_seed(seed)
    sensor = 0.4 + 1.2 * torch.rand(H, W, 1, generator=g)
    sensor[torch.rand(H, W, 1, generator=g) < 0.15] = 0.0
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    mask = (((xx - W / 2) / (0.45 * W)) ** 2 + ((yy - H / 2) / (0.45 * H)) ** 2 < 1.0)[..., None]
    return {"image": torch.rand(H, W, 3, generator=g), "sensor_depth": sensor,
            "normal": torch.rand(H, W, 3, generator=g), "mask": mask}


def test_config3_training_step_matches_oracle(dev):
    """One full training step of config #3 at reduced size through SplatTrainer's tape-free step — get_outputs with
    touch anchors (detached means / opacities / scales), get_loss_dict with mask, sensor-depth holes, mono normals,
    touch-normal MSE — against the CPU oracle (oracle/fusion_ref + oracle/loss_ref): the loss and every parameter
    gradient; and the autograd route (fused node + loss node on the tape) gives the same gradients."""
    from fusionsense_amd.losses import LossConfig
    from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig
    from fusionsense_amd.trainer import PARAM_ORDER, SplatTrainer
    from oracle import loss_ref
    from oracle.fusion_ref import render_fusionsense as render_ref
    W, H = 160, 96
    params, cams = _small_fusionsense(dev, W, H)
    N = params["means"].shape[0]
    n_touch = 150
    g = torch.Generator().manual_seed(3)
    add_mask = torch.zeros(N, dtype=torch.bool)
    add_mask[-n_touch:] = True
    touch_normals = torch.nn.functional.normalize(torch.randn(n_touch, 3, generator=g), dim=-1)
    batch = _synthetic_batch(H, W, 11)
    cam = cams[2]
    cfg = LossConfig()

    def make(direct):
        st = DensifyStrategy(SplatfactoConfig(), num_train_data=9, stats_only=True)
        st.add_mask = add_mask.to(dev)
        st.touch_normals = touch_normals.to(dev)
        tr = SplatTrainer(params, dev, sh_degree=3, strategy=st, loss_cfg=cfg, direct=direct)
        return tr

    tr = make(True)
    fb = _fb(batch, dev, cfg)
    loss, out = tr.train_step(cam, fb, optimizer_step=False)
    grads = {k: tr.slab.views[k].clone() for k in PARAM_ORDER}
    # oracle
    cp = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = render_ref(cp, cam, sh_degree=3, add_mask=add_mask)
    ld = loss_ref.get_loss_dict(ref, dict(batch), cp["scales"], ref["normals_world"], add_mask, touch_normals,
                                loss_ref.LossConfig())
    l_ref = ld["main_loss"] + ld["scale_reg"]
    l_ref.backward()
    assert abs(float(loss) - float(l_ref)) < 2e-5 * abs(float(l_ref)), (float(loss), float(l_ref))
    for k in ("rgb", "depth", "normal"):
        assert rel_err(out[k], ref[k]) < 2e-4, k
    for k in PARAM_ORDER:
        r = cp[k].grad
        # per tensor, relative to that tensor's own scale
        assert rel_err(grads[k], r) < 3e-3, (k, rel_err(grads[k], r))
    # touch anchors: no gradient to means / opacities from the render; their scales still feel the min-scale term
    assert float(grads["means"][-n_touch:].abs().max()) == 0.0 and float(grads["opacities"][-n_touch:].abs().max()) == 0.0
    assert float(grads["scales"][-n_touch:].abs().max()) > 0.0
    # the same step on the autograd tape
    tr2 = make(False)
    loss2, _ = tr2.train_step(cam, fb, optimizer_step=False)
    assert abs(float(loss2) - float(loss)) < 1e-6 * abs(float(loss))
    for k in PARAM_ORDER:
        assert rel_err(tr2.slab.views[k], grads[k]) < 1e-3, k
