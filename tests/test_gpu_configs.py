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


@pytest.mark.parametrize("case", ["full", "nomask", "dense", "mono", "monoonly",
                                  "ndepth", "ndepth_nomask", "eatv", "cosine", "cosine_depth", "l1", "logl1", "mse"])
def test_fusion_loss_kernels_match_reference_goldens(dev, case):
    """ops._FusionLoss against numbers the reference's get_loss_dict produced itself: the gradient images of depth
    and normals, the log-scale gradient, and the value of everything FusionSense adds to the photometric term —
    FusionSense's own configuration and (round 4) the switches it leaves off: normals from depth, EdgeAwareTV, the
    cosine term, the L1 / LogL1 / MSE depth-loss types."""
    from fusionsense_amd import ops
    from fusionsense_amd.losses import LossConfig
    from oracle import loss_ref
    from test_oracle import loss_case_config
    d = np.load(GOLD)
    t = lambda k: torch.from_numpy(d[f"{case}.{k}"])  # noqa: E731
    cfg, intr = loss_case_config(d, case, LossConfig)
    batch = {"image": t("image"), "sensor_depth": t("sensor_depth"), "normal": t("normal_gt")}
    if f"{case}.mono_depth" in d.files:  # the monocular-depth branch (dn_model.py:737-750)
        batch["mono_depth"] = t("mono_depth")
        assert abs(cfg.mono_depth_lambda - float(d[f"{case}.mono_depth_lambda"])) < 1e-7
        if bool(d[f"{case}.mono_only"]):
            del batch["sensor_depth"]
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
    loss = ops.fusion_loss(out, fb, cfg, scales, touch_idx, touch_normals, intr)
    loss.backward()
    # the photometric share of the value comes from the oracle (it does not touch depth / normal / scales)
    photo = loss_ref.photometric_loss(t("rgb").double(), t("image").double(),
                                      t("mask").double() if "mask" in batch else None, cfg.ssim_lambda)
    assert abs(float(loss) - float(photo) - float(d[f"{case}.loss"])) < 2e-6 * max(1.0, abs(float(d[f"{case}.loss"])))
    for name, g in (("v_depth", depth.grad), ("v_normal", normal.grad), ("v_scales", scales.grad)):
        ref = t(name)
        assert torch.allclose(g.cpu(), ref, rtol=2e-5, atol=1e-9), (name, float((g.cpu() - ref).abs().max()))
    if intr is not None:
        # the tape-free route (seed known in the forward: the gradient images leave the forward's own pass, the cosine
        # term accumulates onto them) gives the same numbers
        from fusionsense_amd.fused import _DirectCtx
        one = torch.ones((), device=dev)
        with torch.no_grad():
            lctx = _DirectCtx((True, True, True, False) + (False,) * 8)
            l2 = ops._FusionLoss.forward(lctx, rgb.detach(), depth.detach(), normal.detach(), scales.detach(), fb, cfg,
                                         out["normals_world"], touch_idx, touch_normals, one, True, intr)
            v = ops._FusionLoss.backward(lctx, one)
        assert abs(float(l2) - float(loss)) <= 1e-6 * abs(float(loss))
        assert torch.allclose(v[1], depth.grad, rtol=1e-6, atol=1e-10) and torch.allclose(v[2], normal.grad, rtol=1e-6, atol=1e-10)


def test_pseudo_normals_and_valid_counts(dev):
    """fsgs_normals_from_depth against the oracle's restatement of utils/normal_utils.py (itself pinned through the
    ndepth cases), on a ragged image with a mask; fsgs_depth_valid_counts' three counts against torch."""
    from fusionsense_amd import ops
    from fusionsense_amd._lib import load, ptr, stream_ptr
    from oracle import loss_ref
    g = torch.Generator().manual_seed(5)
    H, W = 37, 53
    depth = 0.5 + 2 * torch.rand(H, W, 1, generator=g)
    mask = (torch.rand(H, W, 1, generator=g) > 0.2).float()
    ref = loss_ref.normal_from_depth((depth * mask).double(), 41.0, 39.0, 25.5, 19.0)
    ref = (1 + ref * torch.tensor([1.0, -1.0, -1.0], dtype=torch.float64)) / 2
    out = torch.empty(H, W, 3, device=dev)
    d_, m_ = depth.to(dev).contiguous(), mask.reshape(H, W).to(dev).contiguous()
    assert load().fsgs_normals_from_depth(H, W, ptr(d_), ptr(m_), 41.0, 39.0, 25.5, 19.0, ptr(out), stream_ptr(dev)) == 0
    # (pixels whose cross product is ~0 — a masked neighbourhood — normalise noise: compare where it is well defined)
    good = (ref - 0.5).abs().sum(-1) > 1e-3
    assert float((out.cpu().double() - ref)[good].abs().max()) < 2e-4 and float(good.float().mean()) > 0.5
    assert torch.equal(out.cpu()[0], torch.full((W, 3), 0.5)) and torch.equal(out.cpu()[:, -1], torch.full((H, 3), 0.5))
    sensor = depth.reshape(H, W) * (torch.rand(H, W, generator=g) > 0.3)
    cx, cy, ca = ops.depth_valid_counts(sensor.to(dev).contiguous(), m_, 0.1)
    valid = (sensor * mask.reshape(H, W)) > 0.1
    assert (cx, cy, ca) == (int(valid[:, :-1].sum()), int(valid[:-1].sum()), int(valid.sum()))


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


# ---------------------------------------------------------------------------------------------------------------
# config #3 at FULL size (1280x720, 80x45 tiles, 60k seeds, 9 views, 5 touch patches): size-independent properties
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def config3(dev):
    from fusionsense_amd.losses import LossConfig
    params, cams = scenes.fusionsense_like_scene(seed=0)
    g = torch.Generator().manual_seed(5)
    params["opacities"] = 1.5 * torch.randn(params["opacities"].shape, generator=g)
    params["features_rest"] = 0.05 * torch.randn(params["features_rest"].shape, generator=g)
    cfg = LossConfig()
    batches = [_fb(_synthetic_batch(720, 1280, 100 + i), dev, cfg) for i in range(3)]
    return params, cams, batches, cfg


def _config3_trainer(dev, params, cfg, direct=True, stats_only=False, patches=None, n_views=9):
    from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig
    from fusionsense_amd.trainer import SplatTrainer
    st = DensifyStrategy(SplatfactoConfig(), num_train_data=n_views, stats_only=stats_only)
    if patches is not None:
        st.set_metadata(touch_patches=patches, gel_scale_factor=6.34e-5, add_touch_at=1000,
                        visual_hull=params["means"][:20_000].clone(), scale_factor=1.0)
    return SplatTrainer(params, dev, sh_degree=3, strategy=st, loss_cfg=cfg, direct=direct, sh_degree_interval=1000), st


def test_config3_full_size_binning_forward_and_step(dev, config3):
    """1280x720 (80x45 tiles, the last tile row 8 pixels high): sorted / consistent lists, a bit-reproducible forward,
    and the tape-free training step against the same step on the autograd tape (loss and every slab gradient)."""
    from fusionsense_amd import ops
    from fusionsense_amd.trainer import PARAM_ORDER
    params, cams, batches, cfg = config3
    tr, _ = _config3_trainer(dev, params, cfg, stats_only=True)
    with torch.no_grad():
        a, b = tr.forward(cams[0]), tr.forward(cams[0])
    info = a["info"]
    assert a["rgb"].shape == (720, 1280, 3) and info.isect_offsets.shape == (1, 45, 80)
    for k in ("rgb", "depth", "normal", "accumulation"):
        assert torch.equal(a[k], b[k]) and bool(torch.isfinite(a[k]).all()), k
    opac = torch.sigmoid(tr.params["opacities"].data).view(1, -1).contiguous()
    tpg, ids, pay, offs = ops.bin_and_sort_live(info.means2d, info.radii, info.depths, info.conics, opac, 80, 45)
    assert torch.equal(pay, info.payload) and torch.equal(offs, info.isect_offsets)
    assert bool((ids[1:] >= ids[:-1]).all())
    tile = (ids >> 32) & ((1 << ops.tile_bits(80 * 45)) - 1)
    bounds = torch.searchsorted(tile.contiguous(), torch.arange(80 * 45 + 1, device=dev))
    assert torch.equal(bounds[:-1].to(torch.int32), offs.flatten()) and int(bounds[-1]) == ids.numel()
    same = ids[1:] == ids[:-1]
    gid = (pay & 0x0FFFFFFF).long()
    assert bool((gid[1:][same] > gid[:-1][same]).all())
    # tape-free step == autograd step
    tr.step = 2100  # SH degree 2, binary opacities active
    la, _ = tr.train_step(cams[1], batches[1], optimizer_step=False)
    ga = {k: tr.slab.views[k].clone() for k in PARAM_ORDER}
    tr2, _ = _config3_trainer(dev, params, cfg, direct=False, stats_only=True)
    tr2.step = 2100
    lb, _ = tr2.train_step(cams[1], batches[1], optimizer_step=False)
    assert abs(float(la) - float(lb)) < 2e-6 * abs(float(lb))
    for k in PARAM_ORDER:
        assert rel_err(ga[k], tr2.slab.views[k]) < 1e-3, k
        assert bool(torch.isfinite(ga[k]).all())
    assert torch.equal(tr.params["opacities"].data, tr2.params["opacities"].data)
    assert set(torch.unique(tr.params["opacities"].data).tolist()) <= {0.0, 1.0}, "binary opacity write (step > warm-up)"


def test_config3_mask_gates_every_image_gradient(dev, config3):
    """With an all-zero object mask the photometric, depth and normal terms see zeros on both sides: the only
    gradient left is the min-scale term's, which reaches the log-scales alone (dn_model.py:702-714, 817-819)."""
    from fusionsense_amd.losses import FrameBatch
    from fusionsense_amd.trainer import PARAM_ORDER
    params, cams, batches, cfg = config3
    fb = batches[0]
    zero = FrameBatch(fb.image, fb.sensor_depth, fb.normal, torch.zeros_like(fb.mask), 1, 1, True)
    tr, _ = _config3_trainer(dev, params, cfg, stats_only=True)
    loss, _ = tr.train_step(cams[0], zero, optimizer_step=False)
    for k in PARAM_ORDER:
        g = tr.slab.views[k]
        if k == "scales":
            assert int((g != 0).sum(dim=-1).max()) == 1 and float(g.min()) >= 0.0 and float(g.max(dim=-1).values.min()) > 0.0, \
                "one positive entry per Gaussian"
        else:
            assert float(g.abs().max()) == 0.0, k
    n = tr.num_gaussians()
    expect = cfg.ssim_lambda * (1 - 1.0) + cfg.normal_lambda * float(torch.exp(tr.params["scales"].data).min(dim=1).values.mean())
    assert abs(float(loss) - expect) < 1e-5 * max(expect, 1e-3), (float(loss), expect, n)


def test_config3_schedule_window_touch_patches_and_densify(dev, config3):
    """Steps 995..1104 of the real schedule at full size: add_touch_patch at step 1000 (BEFORE_TRAIN_ITERATION),
    refinement_after + hull_pruning + touch_pruning at 1000 and 1100, binary opacities, SH degree 0 -> 1.  The anchors
    stay where the patches put them, every step's loss is finite, N follows the reports, and the trainer's state stays
    consistent (slab views, Adam moments, statistics sized for the new N)."""
    from fusionsense_amd.trainer import PARAM_ORDER
    params, cams, batches, cfg = config3
    patches = scenes.touch_patches_on_blob(5, 2000, radius=0.1, seed=0)
    n_touch = sum(p["points_xyz"].shape[0] for p in patches)
    tr, st = _config3_trainer(dev, params, cfg, patches=patches)
    tr.step = 995
    n_hist, losses = [], []
    for it in range(110):
        loss, out = tr.train_step(cams[it % 9], batches[it % 3])
        losses.append(loss)
        n_hist.append(tr.num_gaussians())
        if tr.step == 1001:  # the step-1000 iteration has run: patches were added before it, refinement after it
            assert st.add_mask is not None and int(st.add_mask.sum()) == n_touch
            rep = dict(st.last_report)
            assert rep["step"] == 1000 and rep["n_split"] + rep["n_dup"] > 0, rep
    lv = torch.stack(losses).cpu()
    assert bool(torch.isfinite(lv).all())
    assert int(st.add_mask.sum()) == n_touch and st.add_mask.shape[0] == tr.num_gaussians()
    rows = st.touch_rows()
    pts = torch.cat([p["points_xyz"] for p in patches]).to(dev)
    assert torch.equal(tr.params["means"].data[rows], pts), "touch anchors never move (detached, zero-gradient Adam rows)"
    assert st.last_report["step"] == 1100 and len(set(n_hist)) >= 3, (st.last_report, sorted(set(n_hist)))
    for k in PARAM_ORDER:
        p = tr.params[k]
        # (the window's last step applied Adam inside the backward: no gradients written, .grad says so)
        assert (p.grad is None and not tr.last_step_grads_valid) or p.grad is tr.slab.views[k]
        assert p.shape[0] == tr.num_gaussians() and tr.slab.views[k].shape == p.shape
        s_ = tr.optimizers[k].state[p]
        assert s_["exp_avg"].shape == p.shape and s_["exp_avg_sq"].shape == p.shape
    # the touch-normal MSE is part of the reported loss once the anchors exist (dn_model.py:893-902)
    nw = out["normals_world"][rows]
    mse = float(((nw - st.touch_normals) ** 2).mean())
    assert mse > 0.0


# ---------------------------------------------------------------------------------------------------------------
# config #4: 6 M Gaussians at 1920x1080 (120x68 tiles) — size-independent properties on one GPU, 2 ranks scaled down
# ---------------------------------------------------------------------------------------------------------------
N4, W4, H4 = 6_000_000, 1920, 1080


@pytest.fixture(scope="module")
def config4(dev):
    params = scenes.bicycle_like_scene(N4, seed=0)
    cams = []
    for az in (0.3, 2.1):
        eye = torch.tensor([2.4 * math.cos(az), 2.4 * math.sin(az), 0.9])
        cams.append(scenes.Camera(scenes.look_at_c2w(eye, torch.zeros(3)), 1500.0, 1500.0, W4 / 2.0, H4 / 2.0, W4, H4))
    return {k: v.to(dev) for k, v in params.items()}, cams


def test_config4_full_size_binning_and_forward(dev, config4):
    """6 M Gaussians at 1080p: the frame's lists are sorted by (tile, depth, id) with offsets at the tile boundaries
    (whichever binning route the density picks), the forward is bit-reproducible and in range, last_ids index the
    pixel's own tile list."""
    from fusionsense_amd import ops
    from fusionsense_amd.fused import render_fusionsense_fused
    params, cams = config4
    with torch.no_grad():
        a = render_fusionsense_fused(params, cams[0], sh_degree=3, device=dev)
        b = render_fusionsense_fused(params, cams[0], sh_degree=3, device=dev)
    info = a["info"]
    tw, th = math.ceil(W4 / 16), math.ceil(H4 / 16)
    assert (tw, th) == (120, 68) and info.isect_offsets.shape == (1, th, tw)
    for k in ("rgb", "depth", "normal", "accumulation"):
        assert torch.equal(a[k], b[k]) and bool(torch.isfinite(a[k]).all()), k
    assert torch.equal(info.payload, b["info"].payload) and torch.equal(info.last_ids, b["info"].last_ids)
    assert 0.0 <= float(a["rgb"].min()) and float(a["rgb"].max()) <= 1.0 and float(a["accumulation"].max()) <= 1.0
    pay, offs = info.payload, info.isect_offsets.flatten().long()
    M = pay.numel()
    assert M > 10_000_000 and int((info.radii > 0).sum()) > 1_000_000
    gid = (pay & 0x0FFFFFFF).long()
    assert bool(((pay >> 28) & 0xF != 0).all()) and bool((info.radii[0][gid] > 0).all())
    ends = torch.cat([offs[1:], torch.tensor([M], device=dev)])
    assert bool((ends >= offs).all()) and int(offs[0]) == 0
    # tile of every entry from the offsets; inside a tile: depth non-decreasing, ties in ascending id
    tile_of = torch.searchsorted(offs.contiguous(), torch.arange(M, device=dev), right=True) - 1
    depth_bits = info.depths[0][gid].view(torch.int32).long()
    same_tile = tile_of[1:] == tile_of[:-1]
    assert bool((depth_bits[1:] >= depth_bits[:-1])[same_tile].all())
    tie = same_tile & (depth_bits[1:] == depth_bits[:-1])
    assert bool((gid[1:] > gid[:-1])[tie].all())
    # every entry's Gaussian really overlaps its tile's rectangle
    tx, ty = tile_of % tw, tile_of // tw
    mx, my, r = info.means2d[0][gid, 0], info.means2d[0][gid, 1], info.radii[0][gid].float()
    assert bool(((mx + r > tx * 16) & (mx - r < tx * 16 + 16) & (my + r > ty * 16) & (my - r < ty * 16 + 16)).all())
    # and the independent radix route over the same projected Gaussians gives the same lists, bit for bit
    opac = torch.sigmoid(params["opacities"]).view(1, -1).contiguous()
    st = ops.isect_count_live_async(info.means2d, info.radii, info.conics, opac, tw, th)
    ops.USE_TILE_SORT = False
    try:
        _, _, r_pay, r_offs = ops.isect_finish_live(st, info.means2d, info.radii, info.depths, info.conics, opac, tw, th)
    finally:
        ops.USE_TILE_SORT = True
    assert torch.equal(r_pay, pay) and torch.equal(r_offs, info.isect_offsets)
    last = a["info"].last_ids[0].long()
    yy = torch.arange(H4, device=dev)[:, None] // 16
    xx = torch.arange(W4, device=dev)[None, :] // 16
    t = yy * tw + xx
    hit = a["accumulation"][..., 0] > 0
    assert bool(((last >= offs[t]) & (last < ends[t]))[hit].all())


def test_config4_backward_linear_and_steps_allocation_free(dev, config4):
    """The VJP at 6 M Gaussians is linear in the output gradient and reproducible to the atomics' reordering; a
    training step (forward, loss, backward, Adam over 354 M parameter floats, statistics) allocates no device memory
    once the workspace pool has seen the view."""
    from fusionsense_amd.fused import render_fusionsense_fused
    from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig
    from fusionsense_amd.trainer import SplatTrainer
    params, cams = config4
    g = torch.Generator().manual_seed(1)
    v = [{k: torch.randn(s, generator=g).to(dev) for k, s in (("rgb", (H4, W4, 3)), ("depth", (H4, W4, 1)),
                                                               ("normal", (H4, W4, 3)))} for _ in range(2)]

    def grads(weights):
        p = {k: t.clone().requires_grad_(True) for k, t in params.items()}
        out = render_fusionsense_fused(p, cams[1], sh_degree=3, device=dev)
        torch.autograd.backward([out[k] for k in ("rgb", "depth", "normal")],
                                [sum(w * vi[k] for w, vi in zip(weights, v)) for k in ("rgb", "depth", "normal")])
        return {k: t.grad for k, t in p.items()}

    g1, g2, g12, g12b = grads((1.0, 0.0)), grads((0.0, 1.0)), grads((2.0, 1.0)), grads((2.0, 1.0))
    for k in g12:
        assert bool(torch.isfinite(g12[k]).all())
        assert rel_err(g12[k], 2.0 * g1[k] + g2[k]) < 2e-3, k
        assert rel_err(g12b[k], g12[k]) < 5e-4, k
    del g1, g2, g12, g12b
    st = DensifyStrategy(SplatfactoConfig(), num_train_data=2, stats_only=True)
    tr = SplatTrainer(params, dev, sh_degree=3, strategy=st)
    with torch.no_grad():
        tgt = [{k: o[k].clone() for k in ("rgb", "depth", "normal")} for o in
               (render_fusionsense_fused(params, c, sh_degree=3, device=dev) for c in cams)]
    for c, t in zip(cams, tgt):
        tr.train_step(c, t, optimizer_step=False)
    tr.train_step(cams[0], tgt[0])
    torch.cuda.synchronize()
    n0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    p0 = tr.params["means"].data.clone()
    for s in range(4):
        loss, _ = tr.train_step(cams[s % 2], tgt[s % 2])
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats(dev).get("num_device_alloc", 0) == n0 + 0 or \
        torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - n0 <= 1, "a step must not allocate device memory"
    assert math.isfinite(float(loss)) and not torch.equal(p0, tr.params["means"].data)
    assert st.vis_counts is not None and float(st.vis_counts.max()) >= 2.0


@pytest.mark.parametrize("sharded", ["0", "1"])
def test_config4_two_ranks_scaled_down(sharded):
    """Config #4's data-parallel layout (view-sharded, one 1080p view per rank per step, geometry all-reduce + SH
    factors all-gather) with 2 ranks sharing this GPU over gloo, 200 k Gaussians: bench.py as the driver launches it;
    the ranks end with bit-identical replicas and the line carries the exchange's size."""
    from helpers import run_bench_ranks
    line, err = run_bench_ranks(2, ["--config", "4", "--n-gauss", "200000", "--views", "4", "--steps", "4", "--warmup", "1",
                                    "--no-cpu-baseline"],
                                dict(FSGS_DIST_BACKEND="gloo", FSGS_BENCH_CHECK_REPLICAS="1", FSGS_SHARDED_GEOMETRY=sharded))
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["sharded_geometry"] is (sharded == "1")
    assert line["config"]["baseline_config"] == 4 and line["config"]["width"] == 1920
    assert line["config"]["comm_bytes_per_step_per_rank"] > 0 and line["config"]["backend"] == "gloo"
    assert err.count("replicas identical") == 2, err[-2000:]


def test_bench_spawns_its_own_ranks_without_a_launcher():
    """``python bench.py --gpus 2`` with NO launcher (the way the driver starts the 1-GPU run): bench.py starts its two
    ranks itself as a child torch.distributed.run before touching the GPU (here both on this GPU over gloo), relays
    rank 0's line and reports what the group really was — n_gpus 2, both ranks seen, replicas checked by default."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(FSGS_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--n-gauss", "20000", "--res", "160", "--views", "4", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=400, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    c = line["config"]
    assert c["world_size"] == 2 and sorted(c["ranks_seen"]) == [0, 1] and c["backend"] == "gloo"
    assert c["replicas_identical_after_timed_steps"] is True and c["launched_by"].startswith("bench.py")
    assert c["parallelism"] == "dp2" and c["comm_bytes_per_step_per_rank"] > 0
    # the same flags over RCCL on a box with one GPU: a loud refusal, not a one-rank measurement
    if torch.cuda.device_count() < 2:
        env.pop("FSGS_DIST_BACKEND")
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, env=env, timeout=200, cwd=root)
        assert r.returncode == 2 and not r.stdout.strip()


@pytest.mark.parametrize("sharded", ["0", "1"])
def test_bench_with_eight_ranks_over_gloo_on_this_gpu(sharded):
    """``python bench.py --gpus 8`` — the command of the driver's scaling run — with all EIGHT ranks sharing this GPU over
    gloo (functional, small N; VERDICT r3 Next 3: nothing had ever run at W = 8): spawned by bench.py itself under its
    watchdog, the 8-way SH-factor all-gather + in-order rebuild, the geometry half all-reduced or reduce-scattered into
    eight padded shards (N = 20 011 is no multiple of 64 x 8), replicas bit-identical, and the line carries the
    per-rank step time and the exposed communication time a scaling curve is read with."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(FSGS_DIST_BACKEND="gloo", FSGS_SHARDED_GEOMETRY=sharded, FSGS_BENCH_VERBOSE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1",
                        "--n-gauss", "20011", "--res", "160", "--views", "16", "--no-cpu-baseline", "--spawn-timeout", "500"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    c = line["config"]
    assert line["n_gpus"] == 8 and c["world_size"] == 8 and sorted(c["ranks_seen"]) == list(range(8))
    assert c["parallelism"] == "dp8" and c["backend"] == "gloo" and c["launched_by"].startswith("bench.py")
    assert c["replicas_identical_after_timed_steps"] is True and r.stderr.count("replicas identical") == 8
    assert c["sharded_geometry"] is (sharded == "1") and c["comm_bytes_per_step_per_rank"] > 0
    sm, cm = line["step_ms_per_rank"], line["comm_ms_exposed"]
    assert 0 < sm["min"] <= sm["median"] <= sm["max"] and 0 <= cm["min"] <= cm["max"] < sm["max"]
    assert line["value"] == pytest.approx(8 * 4 / (line["ms_per_step"] * 4e-3), rel=1e-3)  # whole-job views per second


def test_two_ranks_over_rccl_when_two_gpus():
    """The first thing a real multi-GPU node runs: bench.py with 2 ranks on 2 GPUs over RCCL ("nccl"), replicas
    checked bit for bit.  Skipped on 1-GPU boxes (the driver's 8-GPU scaling run is the measurement)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    from helpers import run_bench_ranks
    line, err = run_bench_ranks(2, ["--steps", "10", "--warmup", "2", "--n-gauss", "100000", "--res", "400", "--views", "8",
                                    "--no-cpu-baseline"], dict(FSGS_BENCH_CHECK_REPLICAS="1"))
    assert line["n_gpus"] == 2 and line["config"]["backend"] == "nccl" and line["value"] > 0
    assert err.count("replicas identical") == 2, err[-2000:]


def test_rccl_collectives_of_the_multi_gpu_step_in_a_one_rank_group():
    """The RCCL calls of the data-parallel step (all_reduce AVG on the geometry half, asynchronous
    all_gather_into_tensor of the SH factors, broadcast / barrier / MAX of the bench) really issued on this box's one
    GPU: a one-rank "nccl" group with FSGS_FORCE_COLLECTIVES=1 (identities, but the same library calls, stream
    ordering and deferred feature update as with 8 ranks).  The training must equal the plain single-process run."""
    from helpers import run_bench_ranks
    args = ["--steps", "8", "--warmup", "2", "--n-gauss", "50000", "--res", "320", "--views", "4", "--no-cpu-baseline"]
    line, err = run_bench_ranks(1, args, dict(FSGS_FORCE_COLLECTIVES="1", FSGS_FORCE_SPLIT_STEP="1",
                                              FSGS_BENCH_CHECK_REPLICAS="1", FSGS_BENCH_PARAM_DIGEST="1"))
    assert line["n_gpus"] == 1 and line["config"]["backend"] == "nccl" and line["value"] > 0
    assert line["config"]["comm_bytes_per_step_per_rank"] > 0, "the collectives were not issued"
    assert "replicas identical" in err
    plain, err2 = run_bench_ranks(1, args, dict(FSGS_FORCE_SPLIT_STEP="1", FSGS_BENCH_PARAM_DIGEST="1"))
    assert plain["config"]["backend"] is None
    import json
    dig = lambda e: json.loads([l for l in e.splitlines() if "param digest" in l][-1].split("param digest")[1])
    a, b = dig(err), dig(err2)
    for name in a:  # (float atomics: equal to rounding; a missing or doubled update would move these by percents)
        assert abs(a[name][1] - b[name][1]) <= 1e-4 * abs(b[name][1]) + 1e-7, (name, a[name], b[name])
        assert abs(a[name][0] - b[name][0]) <= 1e-4 * abs(b[name][1]) + 1e-7, (name, a[name], b[name])
    # the sharded geometry step's RCCL calls (reduce_scatter_tensor AVG, the in-place all_gather_into_tensor of the
    # parameter slab): identities in a one-rank group, the same training
    shard, err3 = run_bench_ranks(1, args, dict(FSGS_FORCE_COLLECTIVES="1", FSGS_FORCE_SPLIT_STEP="1", FSGS_SHARDED_GEOMETRY="1",
                                                FSGS_BENCH_CHECK_REPLICAS="1", FSGS_BENCH_PARAM_DIGEST="1"))
    assert shard["config"]["backend"] == "nccl" and shard["config"]["sharded_geometry"] is True
    c = dig(err3)
    for name in c:
        assert abs(c[name][1] - b[name][1]) <= 1e-4 * abs(b[name][1]) + 1e-7, (name, c[name], b[name])
        assert abs(c[name][0] - b[name][0]) <= 1e-4 * abs(b[name][1]) + 1e-7, (name, c[name], b[name])


# ---------------------------------------------------------------------------------------------------------------
# inference callers (row N4): eval loop and mesh back-projection over the HIP forward
# ---------------------------------------------------------------------------------------------------------------
def test_eval_loop_and_backprojection(dev):
    """The eval pass (get_outputs_for_camera + metrics + num_rays_per_sec / fps, dn_pipeline.py:233-253) and the mesh
    exporters' back-projection loop (export_mesh.py:338-455) over the HIP forward: rendering a scene against its own
    renders scores perfectly; back-projected points re-project onto their pixels at their rendered depth; the
    normals are unit vectors; a crop box reaches the render."""
    from fusionsense_amd import inference as inf
    from fusionsense_amd.crop import OrientedBox
    from fusionsense_amd.trainer import SplatTrainer
    params, cams = _small_fusionsense(dev, 160, 96)
    params["opacities"] = params["opacities"] + 3.0
    cams = cams[:3]
    tr = SplatTrainer(params, dev, sh_degree=3)
    batches = []
    for c in cams:
        o = inf.get_outputs_for_camera(tr, c)
        batches.append({"image": o["rgb"].clone(), "sensor_depth": o["depth"].clone(), "normal": o["normal"].clone(),
                        "mask": (o["accumulation"] > 0.5)})
    m = inf.eval_loop(tr, cams, batches)
    assert m["rgb_psnr"] == float("inf") and abs(m["rgb_ssim"] - 1.0) < 1e-5
    assert m["depth_abs_rel"] == 0.0 and m["depth_a1"] == 1.0 and m["normal_mean_err"] == 0.0
    assert m["num_rays_per_sec"] > 0 and abs(m["fps"] - m["num_rays_per_sec"] / (160 * 96)) < 1e-6 * m["fps"]
    # a different scene scores worse
    p2, _ = _small_fusionsense(dev, 160, 96, seed=4)
    m2 = inf.eval_loop(SplatTrainer(p2, dev, sh_degree=3), cams, batches, get_std=True)
    assert "rgb_psnr_std" in m2 and m2["rgb_psnr"] < 40 and m2["rgb_ssim"] < 0.99 and m2["depth_abs_rel"] > 0
    # back-projection
    g = torch.Generator().manual_seed(0)
    pts, cols, nrm = inf.backproject_views(tr, cams, batches, total_points=3000, generator=g)
    assert pts.shape == cols.shape == nrm.shape and pts.shape[0] > 2000
    assert torch.allclose(nrm.norm(dim=-1), torch.ones_like(nrm[:, 0]), atol=1e-4)
    cam = cams[0]
    out = inf.get_outputs_for_camera(tr, cam)
    c2w = torch.eye(4, device=dev)
    c2w[:3, :4] = cam.c2w.to(dev)
    c2w = (c2w @ torch.diag(torch.tensor([1.0, -1.0, -1.0, 1.0], device=dev)))[:3, :4]
    P, _ = inf.get_means3d_backproj(out["depth"], cam.fx, cam.fy, cam.cx, cam.cy, (160, 96), c2w)
    uvz = inf.project_pix(P, cam.fx, cam.fy, cam.cx, cam.cy, c2w, return_z_depths=True)
    coords = inf.get_camera_coords((160, 96), device=dev)
    assert torch.allclose(uvz[:, :2], coords, atol=2e-3) and torch.allclose(uvz[:, 2], out["depth"].reshape(-1), rtol=1e-5)
    # world points of the object blob lie near the blob (radius 0.1 around the origin) where the blob is hit
    near = out["depth"].reshape(-1) < 1.2
    assert float(P[near].norm(dim=-1).median()) < 0.6
    # obb crop: only the blob survives, the image changes
    box = OrientedBox(torch.eye(3), torch.zeros(3), torch.full((3,), 0.25))
    cropped = inf.get_outputs_for_camera(tr, cam, obb_box=box)
    assert cropped["rgb"].shape == out["rgb"].shape and not torch.equal(cropped["rgb"], out["rgb"])


def test_eval_loop_scores_equal_the_oracle_renders_scores(dev):
    """The eval pass over the HIP forward against the SAME pass over the CPU oracle's renders (the eval loop
    had only been compared with itself): a cube scene scored against supervision rendered by the oracle from another
    scene — every metric of dn_model.py:1032-1148 / metrics.py agrees to the render tolerance."""
    from fusionsense_amd import inference as inf
    from fusionsense_amd.trainer import SplatTrainer
    from oracle.fusion_ref import render_fusionsense as render_ref
    params, cam = scenes.cube_scene(1500, seed=3)
    other, _ = scenes.cube_scene(1500, seed=8)
    with torch.no_grad():
        sup = render_ref({k: v.clone() for k, v in other.items()}, cam, sh_degree=3)
        ref = render_ref({k: v.clone() for k, v in params.items()}, cam, sh_degree=3)
    batch = {"image": sup["rgb"], "sensor_depth": sup["depth"], "normal": sup["normal"], "mask": sup["accumulation"] > 0.3}
    m_ref, _ = inf.get_image_metrics_and_images({k: ref[k] for k in ("rgb", "depth", "normal")}, batch)
    tr = SplatTrainer(params, dev, sh_degree=3)
    out = inf.get_outputs_for_camera(tr, cam)
    for k in ("rgb", "depth", "normal", "accumulation"):
        assert rel_err(out[k], ref[k]) < 1e-4, k
    m = inf.eval_loop(tr, [cam], [batch])
    for k, v in m_ref.items():
        if k == "rgb_lpips":
            continue
        assert abs(m[k] - v) <= 2e-4 * max(1.0, abs(v)), (k, m[k], v)
    assert 5 < m["rgb_psnr"] < 40 and m["depth_abs_rel"] > 0  # (a real comparison, not a scene against itself)


@pytest.mark.parametrize("case", ["closest", "analytical"])
def test_level_set_extraction_over_the_hip_forward(dev, case):
    """Row N4's last piece: ``compute_level_surface_points`` (dn_model.py:1706-1946) over the HIP forward against the
    REFERENCE's own method executed over the oracle rasterizer (tests/golden/reference_levelset.npz): the same pixels
    cross each level (a handful sit on a threshold of the 1e-4-different depth images), at the same points, with the same
    normals; then the LevelSetExtractor loop (export_mesh.py:506-583) over several cameras."""
    import numpy as np
    from fusionsense_amd import inference as inf
    from fusionsense_amd.trainer import SplatTrainer
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_levelset.npz"))
    t = lambda k: torch.from_numpy(d[k])  # noqa: E731
    params = {k: t(f"in.{k}") for k in ("means", "scales", "quats", "features_dc", "features_rest", "opacities")}
    fx, fy, cx, cy, W, H = (float(x) for x in d["intr"])
    W, H = int(W), int(H)
    cam = scenes.Camera(t("c2w"), fx, fy, cx, cy, W, H)
    tr = SplatTrainer(params, dev, sh_degree=3, sh_degree_interval=1000)
    tr.step = 100  # (the golden's model was at step 100: sh_degree_to_use = 100 // 1000 = 0, dn_model.py:562-565)
    o = tr.forward(cam)
    assert rel_err(o["depth"], t("render.depth")) < 1e-4 and rel_err(o["normals_world"], t("render.normals_world")) < 1e-5
    assert rel_err(o["rgb"], t("render.rgb")) < 1e-4
    mode = ["closest_gaussian", "analytical"][int(d[f"{case}.mode"])]
    mask = t("mask") if bool(d[f"{case}.use_mask"]) else None
    res = inf.compute_level_surface_points(tr, cam, 10 ** 9, mask, (0.1, 0.3, 0.5), mode,
                                           sample_fn=lambda n, k: list(range(n))[:k])
    c2w = torch.eye(4)
    c2w[:3] = cam.c2w
    c2w = (c2w @ torch.diag(torch.tensor([1.0, -1.0, -1.0, 1.0])))[:3]
    for lv in (0.1, 0.3, 0.5):
        ref_p, ref_n, ref_c = (t(f"{case}.{lv}.{k}") for k in ("points", "normals", "colors"))
        uv = inf.project_pix(ref_p, fx, fy, cx, cy, c2w)  # the pixel every reference point came from
        ref_pid = (torch.round(uv[:, 1] - 0.5).long() * W + torch.round(uv[:, 0] - 0.5).long())
        pid = res[lv]["pixel_ids"].cpu()
        common, ia, ib = np.intersect1d(pid.numpy(), ref_pid.numpy(), return_indices=True)
        assert len(common) >= 0.98 * max(len(pid), len(ref_pid)), (lv, len(pid), len(ref_pid), len(common))
        got_p, got_n, got_c = (res[lv][k].cpu()[ia] for k in ("points", "normals", "colors"))
        # (a pixel whose 2nd-nearest Gaussian changes with the 1e-4 depth difference samples another ray interval)
        close = (got_p - ref_p[ib]).norm(dim=-1) < 2e-3
        assert float(close.float().mean()) > 0.97, (lv, float(close.float().mean()))
        ok_n = (got_n[close] - ref_n[ib][close]).abs().max(dim=-1).values < (2e-2 if mode == "analytical" else 1e-3)
        assert float(ok_n.float().mean()) > 0.97, (lv, float(ok_n.float().mean()))
        dc = (got_c[close] - ref_c[ib][close]).abs().max(dim=-1).values  # (rgb: 1e-4 of the image, but for the pixels
        assert float((dc < 5e-4).float().mean()) > 0.99 and float(dc.max()) <= 1.0 / 255.0 + 1e-4, (lv, float(dc.max()))  # on an alpha >= 1/255 decision)
    # the extractor's loop: points of three cameras, per level, sub-sampled per frame
    cams = [cam, scenes.Camera(scenes.look_at_c2w(torch.tensor([2.0, 1.0, 1.2]), torch.zeros(3)), fx, fy, cx, cy, W, H),
            scenes.Camera(scenes.look_at_c2w(torch.tensor([-1.5, 2.0, 0.4]), torch.zeros(3)), fx, fy, cx, cy, W, H)]
    sets = inf.extract_level_sets(tr, cams, total_points=600, return_normal=mode)
    for lv, (P, C, Nn) in sets.items():
        assert P.shape == C.shape == Nn.shape and 0 < P.shape[0] <= 3 * 201
        assert torch.allclose(Nn.norm(dim=-1), torch.ones_like(Nn[:, 0]), atol=1e-3)
        assert float(P.abs().max()) < 1.5  # on / near the unit cube's surface


# ---------------------------------------------------------------------------------------------------------------
# config #5: half-precision attribute storage (features, scales, quats, opacities as IEEE-half mirrors; fp32 master + Adam)
# ---------------------------------------------------------------------------------------------------------------
def _rounded(params):
    from fusionsense_amd.trainer import HALF_GROUPS
    return {k: (v.half().float() if k in HALF_GROUPS else v.clone()) for k, v in params.items()}


def test_half_attribute_step_matches_oracle_on_rounded_inputs(dev):
    """The half-storage path reads fp16 mirrors and computes in fp32: its images and parameter gradients must equal
    the CPU oracle's on the fp16-ROUNDED attributes (means stay fp32) — and be bit-identical to the fp32 HIP path
    fed the rounded values, since only the loads differ."""
    from fusionsense_amd.trainer import PARAM_ORDER, SplatTrainer
    from oracle.fusion_ref import render_fusionsense as render_ref
    params, cam = scenes.cube_scene(3000, seed=4)
    g = torch.Generator().manual_seed(2)
    tgt = {"rgb": torch.rand(128, 128, 3, generator=g).to(dev), "depth": torch.rand(128, 128, 1, generator=g).to(dev) + 1.5,
           "normal": torch.rand(128, 128, 3, generator=g).to(dev)}
    h = SplatTrainer(params, dev, sh_degree=3, half_attributes=True)
    lh, oh = h.train_step(cam, tgt, optimizer_step=False)
    f = SplatTrainer(_rounded(params), dev, sh_degree=3)
    lf, of = f.train_step(cam, tgt, optimizer_step=False)
    for k in ("rgb", "depth", "normal", "accumulation"):
        assert torch.equal(oh[k], of[k]), k
    assert torch.equal(oh["info"].payload, of["info"].payload) and torch.equal(oh["info"].radii, of["info"].radii)
    assert abs(float(lh) - float(lf)) <= 1e-6 * abs(float(lf))
    for k in PARAM_ORDER:  # (the backward's float atomics reorder sums between any two runs)
        assert rel_err(h.slab.views[k], f.slab.views[k]) < 5e-4, k
    # against the oracle on the rounded inputs
    cp = {k: v.clone().requires_grad_(True) for k, v in _rounded(params).items()}
    ref = render_ref(cp, cam, sh_degree=3)
    for k in ("rgb", "depth", "normal"):
        assert rel_err(oh[k], ref[k]) < 2e-4, k
    l_ref = (0.8 * (ref["rgb"] - tgt["rgb"].cpu()).abs().mean() + 0.2 * (ref["depth"] - tgt["depth"].cpu()).abs().mean()
             + 0.1 * (ref["normal"] - tgt["normal"].cpu()).abs().mean())
    # (L1 terms only for the oracle-side gradient check: compare through a trainer-independent backward)
    o2 = h.forward(cam)
    l_hip = (0.8 * (o2["rgb"] - tgt["rgb"]).abs().mean() + 0.2 * (o2["depth"] - tgt["depth"]).abs().mean()
             + 0.1 * (o2["normal"] - tgt["normal"]).abs().mean())
    h.slab.zero_()
    l_hip.backward()
    l_ref.backward()
    assert abs(float(l_hip) - float(l_ref)) < 1e-4 * abs(float(l_ref))
    for k in PARAM_ORDER:
        assert rel_err(h.slab.views[k], cp[k].grad) < 3e-3, (k, rel_err(h.slab.views[k], cp[k].grad))


def test_half_mirrors_follow_adam_densify_and_binary_write(dev):
    """The mirrors are rewritten by the fused Adam launch (bit-equal to master.half()), re-derived after densify /
    prune moves rows and after an opacity reset, and the binary-opacity write lands in master and mirror."""
    from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig
    from fusionsense_amd.trainer import HALF_GROUPS, SplatTrainer
    params = scenes.lego_like_scene(8000, seed=3)
    cams = scenes.hemisphere_cameras(2, width=160, height=128, focal=180.0, seed=1)
    g = torch.Generator().manual_seed(0)
    tgts = [{"rgb": torch.rand(128, 160, 3, generator=g).to(dev), "depth": torch.rand(128, 160, 1, generator=g).to(dev),
             "normal": torch.rand(128, 160, 3, generator=g).to(dev)} for _ in cams]
    cfg = SplatfactoConfig(warmup_length=4, refine_every=5, reset_alpha_every=4, stop_split_at=1000, cull_alpha_thresh=0.3)
    st = DensifyStrategy(cfg, num_train_data=2)
    tr = SplatTrainer(params, dev, sh_degree=3, strategy=st, half_attributes=True)
    n_seen = set()
    for it in range(32):
        tr.train_step(cams[it % 2], tgts[it % 2])
        n_seen.add(tr.num_gaussians())
        hm = tr.half_mirrors()
        for k in HALF_GROUPS:
            assert hm[k].dtype == torch.float16 and torch.equal(hm[k], tr.params[k].data.half()), (it, k)
    assert len(n_seen) > 1, "the run must have moved rows"
    # masters keep what the mirrors cannot hold
    assert not torch.equal(tr.params["features_dc"].data, tr.half_mirrors()["features_dc"].float())
    # binary opacities (step > warm-up, outside the post-reset margin is not reachable this early: force the write)
    from fusionsense_amd.fused import render_fusionsense_fused
    with torch.no_grad():
        render_fusionsense_fused(tr.params, cams[0], sh_degree=3, device=dev, binary_threshold=0.9, half=tr.half_mirrors())
    op = tr.params["opacities"].data
    assert set(torch.unique(op).tolist()) <= {0.0, 1.0} and torch.equal(tr.half_mirrors()["opacities"], op.half())


N5 = 10_000_000


def test_config5_full_size_half_storage(dev):
    """10 M Gaussians, SH degree 3, 1920x1080, attributes read from half mirrors: the forward is bit-reproducible and
    equal to rounding to the fp32 path on the rounded values (measured: 2e-6 of the pixels differ by more than 1e-5); a training step is finite, allocation-free once primed,
    and leaves mirrors == master.half(); parameter + optimizer + mirror memory is what the sizing says."""
    from fusionsense_amd.fused import render_fusionsense_fused
    from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig
    from fusionsense_amd.trainer import HALF_GROUPS, SplatTrainer
    params = scenes.bicycle_like_scene(N5, seed=0)
    eye = torch.tensor([2.4 * math.cos(0.3), 2.4 * math.sin(0.3), 0.9])
    cam = scenes.Camera(scenes.look_at_c2w(eye, torch.zeros(3)), 1500.0, 1500.0, W4 / 2.0, H4 / 2.0, W4, H4)
    st = DensifyStrategy(SplatfactoConfig(), num_train_data=1, stats_only=True)
    tr = SplatTrainer(params, dev, sh_degree=3, strategy=st, half_attributes=True)
    hm = tr.half_mirrors()
    assert sum(t.numel() * 2 for t in hm.values()) == N5 * 56 * 2  # 112 B of half attributes (+ 12 B fp32 means = 124)
    with torch.no_grad():
        a, b = tr.forward(cam), tr.forward(cam)
        rp = {k: v.to(dev) for k, v in _rounded(params).items()}
        c = render_fusionsense_fused(rp, cam, sh_degree=3, device=dev)
    for k in ("rgb", "depth", "normal", "accumulation"):
        assert torch.equal(a[k], b[k]) and bool(torch.isfinite(a[k]).all()), k
        # the half instantiation of the projecting count pass is a separately compiled inline of the same statements
        # (FMA contraction may differ): equal to rounding, a handful of pixels on an alpha >= 1/255 decision
        d = (a[k] - c[k]).abs()
        assert float((d > 1e-4 * max(1.0, float(c[k].abs().max()))).float().mean()) < 1e-4, k
    assert abs(a["info"].payload.numel() - c["info"].payload.numel()) <= 1e-5 * c["info"].payload.numel()
    del rp, c
    tgt = {k: a[k].clone() * 0.9 for k in ("rgb", "depth", "normal")}
    tr.train_step(cam, tgt, optimizer_step=False)
    tr.train_step(cam, tgt)
    tr.train_step(cam, tgt)  # (the list capacity and the scratch pool of this process have settled after two real steps)
    torch.cuda.synchronize()
    n0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    for _ in range(3):
        loss, _ = tr.train_step(cam, tgt)
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - n0 <= 1
    assert math.isfinite(float(loss))
    for k in HALF_GROUPS:
        assert torch.equal(tr.half_mirrors()[k], tr.params[k].data.half()), k


# ---------------------------------------------------------------------------------------------------------------
# integration.patch: DNSplatterModel.get_outputs on the fused node, against the reference's own get_outputs
# ---------------------------------------------------------------------------------------------------------------
class _FakeCameras:
    def __init__(self, c2w, fx, fy, cx, cy, W, H):
        self.camera_to_worlds = c2w[None]
        self._k = torch.tensor([[[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]]])
        self.width, self.height = torch.tensor([[W]]), torch.tensor([[H]])
        self.shape = (1,)
        self.metadata = {"cam_idx": 3}

    def rescale_output_resolution(self, f):
        pass

    def get_intrinsics_matrices(self):
        return self._k


@pytest.mark.parametrize("case", ["plain", "binary_anchor"])
def test_patched_get_outputs_matches_reference_execution(dev, case):
    """integration.get_outputs_fused bound to a model object, against what the reference's own get_outputs produced
    on the same parameters (tests/golden/reference_model.npz; rasterizer = the CPU oracle there, the fused HIP node
    here): the returned images, the binary-opacity write on the parameter, the model-side attributes, and the
    parameter gradients with the touch anchors detached."""
    import types
    from fusionsense_amd import integration
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_model.npz"))
    t = lambda k: torch.from_numpy(d[k])  # noqa: E731
    names = ["means", "scales", "quats", "features_dc", "features_rest", "opacities"]
    fx, fy, cx, cy, W, H = [float(v) for v in d["go.intr"]]
    step = int(d[f"go.{case}.step"])

    class Model:
        training = True
        crop_box = None
        add_mask = None
        get_outputs = None

        def _get_downscale_factor(self):
            return 1

        def _get_background_color(self):
            return torch.ones(3)

    integration.patch(Model)
    m = Model()
    m.config = types.SimpleNamespace(use_binary_opacities=True, binary_opacities_threshold=0.9, warmup_length=500,
                                     reset_alpha_every=30, refine_every=100, sh_degree=3, sh_degree_interval=1000,
                                     rasterize_mode="classic", predict_normals=True)
    m.step = step
    m.gauss_params = {k: torch.nn.Parameter(t(f"go.in.{k}").to(dev)) for k in names}
    m.camera_optimizer = types.SimpleNamespace(apply_to_camera=lambda cam: cam.camera_to_worlds)
    if f"go.{case}.add_mask" in d.files:
        m.add_mask = t(f"go.{case}.add_mask").to(dev)
    cam = _FakeCameras(t("go.c2w"), fx, fy, cx, cy, int(W), int(H))
    out = m.get_outputs(cam)
    assert set(out) == {"rgb", "depth", "normal", "accumulation", "background"}
    loss = sum((out[k] * t(f"go.w.{k}").to(dev)).mean() for k in ("rgb", "depth", "normal"))
    loss.backward()
    for k in ("rgb", "depth", "accumulation"):
        assert rel_err(out[k], t(f"go.{case}.out.{k}")) < 2e-4, k
    dn = (out["normal"].detach().cpu() - t(f"go.{case}.out.normal")).abs()
    assert float(dn.mean()) < 1e-4 and float((dn > 1e-2).float().mean()) < 2e-3
    assert torch.equal(m.gauss_params["opacities"].detach().cpu(), t(f"go.{case}.opacities_after"))
    assert torch.allclose(m.gauss_params["normals"].detach().cpu(), t(f"go.{case}.normals_world"), atol=1e-5)
    assert m.last_size == (int(H), int(W)) and m.camera is cam and m.camera_idx == 3
    assert m.radii.shape == (260,) and m.xys.absgrad.shape == (1, 260, 2)
    assert torch.equal(m.vis_indices, torch.where(m.radii > 0)[0])
    for k in names:
        g = m.gauss_params[k].grad
        g = g.cpu() if g is not None else torch.zeros_like(t(f"go.in.{k}"))
        assert rel_err(g, t(f"go.{case}.grad.{k}")) < 3e-3, (k, rel_err(g, t(f"go.{case}.grad.{k}")))


@pytest.mark.parametrize("case", ["plain", "binary_anchor"])
def test_shim_only_route_matches_reference_execution(dev, case):
    """The FIRST integration step (gsplat_shim: the reference's own get_outputs body, op by op, over
    rasterization() + rasterize_gaussians() on libfsgs — fusion.render_fusionsense mirrors that body) against the
    same reference-execution goldens as the patched route: images, binary write, gradients with detached anchors."""
    from fusionsense_amd import integration
    from fusionsense_amd.fusion import render_fusionsense
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_model.npz"))
    t = lambda k: torch.from_numpy(d[k])  # noqa: E731
    names = ["means", "scales", "quats", "features_dc", "features_rest", "opacities"]
    fx, fy, cx, cy, W, H = [float(v) for v in d["go.intr"]]
    step = int(d[f"go.{case}.step"])
    cfg = __import__("types").SimpleNamespace(use_binary_opacities=True, binary_opacities_threshold=0.9, warmup_length=500,
                                              reset_alpha_every=30, refine_every=100)
    gp = {k: torch.nn.Parameter(t(f"go.in.{k}").to(dev)) for k in names}
    add_mask = t(f"go.{case}.add_mask").to(dev) if f"go.{case}.add_mask" in d.files else None
    cam = scenes.Camera(t("go.c2w").float(), fx, fy, cx, cy, int(W), int(H))
    out = render_fusionsense(gp, cam, sh_degree=min(step // 1000, 3), device=dev, add_mask=add_mask,
                             binary_threshold=integration._binary_threshold(cfg, step))
    loss = sum((out[k] * t(f"go.w.{k}").to(dev)).mean() for k in ("rgb", "depth", "normal"))
    loss.backward()
    for k in ("rgb", "depth", "accumulation"):
        assert rel_err(out[k], t(f"go.{case}.out.{k}")) < 2e-4, k
    dn = (out["normal"].detach().cpu() - t(f"go.{case}.out.normal")).abs()
    assert float(dn.mean()) < 1e-4 and float((dn > 1e-2).float().mean()) < 2e-3
    assert torch.equal(gp["opacities"].detach().cpu(), t(f"go.{case}.opacities_after"))
    assert torch.allclose(out["normals_world"].detach().cpu(), t(f"go.{case}.normals_world"), atol=1e-5)
    assert out["xys"].absgrad.shape == (1, 260, 2) and out["radii"].shape == (260,)
    for k in names:
        g = gp[k].grad
        g = g.cpu() if g is not None else torch.zeros_like(t(f"go.in.{k}"))
        assert rel_err(g, t(f"go.{case}.grad.{k}")) < 3e-3, (k, rel_err(g, t(f"go.{case}.grad.{k}")))


def test_scene_directory_to_training_checkpoint_and_export(dev, tmp_path):
    """Row N3 on the GPU: a FusionSense scene directory on disk (transforms.json, images, masks, 16-bit sensor depth,
    mono normals, seed + hull PLYs, gelsight touch patches) -> fusionsense_amd.dataparser -> seed Gaussians
    (populate_modules) -> SplatTrainer steps with the reference's get_loss_dict over prepared FrameBatches, across
    add_touch_patch and one refinement + hull / touch pruning -> nerfstudio checkpoint and ``ns-export gaussian-splat``
    PLY -> reloaded: bit-identical parameters and forward."""
    from fusionsense_amd import dataparser as dp, io as fio
    from fusionsense_amd.fused import render_fusionsense_fused
    from fusionsense_amd.losses import FrameBatch, LossConfig
    from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig
    from fusionsense_amd.trainer import PARAM_ORDER, SplatTrainer
    from helpers import write_scene_dir
    write_scene_dir(tmp_path, np.random.default_rng(3), variant="plain", H=96, W=128, n_pts=3000, n_hull=300, look_at=True)
    out = dp.parse_transforms(tmp_path, load_touches=True)
    md = out.metadata
    assert len(out.cameras) == 3 and md["points3D_xyz"].shape == (3000, 3) and len(md["touch_patches"]) == 2
    gp = dp.seed_gauss_params(out)
    cfg = LossConfig()
    batches = [_fb(dp.load_batch(out, i), dev, cfg) for i in range(len(out.cameras))]
    assert all(isinstance(b, FrameBatch) and b.has_sensor and b.mask is not None and b.normal is not None for b in batches)
    st = DensifyStrategy(SplatfactoConfig(), num_train_data=len(out.cameras))
    st.set_metadata(touch_patches=md["touch_patches"], gel_scale_factor=md["gel_scale_factor"], add_touch_at=595,
                    visual_hull=md["visual_hull"], scale_factor=md["scale_factor"])
    tr = SplatTrainer(gp, dev, sh_degree=3, strategy=st, seed=0, sh_degree_interval=1000, loss_cfg=cfg)
    tr.step = 590  # steps 590 .. 609: add_touch_patch at 595, refinement_after + hull_pruning + touch_pruning at 600
    n0, losses_ = tr.num_gaussians(), []
    for s in range(20):
        loss, o = tr.train_step(out.cameras[s % 3], batches[s % 3])
        losses_.append(float(loss))
    assert all(math.isfinite(v) for v in losses_) and tr.step == 610
    assert st.add_mask is not None and int(st.add_mask.sum()) > 0, "the touch patches were added as anchors"
    assert tr.num_gaussians() != n0 and float(o["accumulation"].max()) > 0.2, "the cameras see the scene"
    # checkpoint + export, reloaded into a fresh trainer / a plain parameter dictionary
    ck, ply = str(tmp_path / "step-000000609.ckpt"), str(tmp_path / "splat.ply")
    fio.save_checkpoint(tr, ck)
    assert fio.export_gaussian_splat_ply(tr.params, ply) == tr.num_gaussians()
    tr2 = SplatTrainer(gp, dev, sh_degree=3, seed=0, sh_degree_interval=1000, loss_cfg=cfg,
                       strategy=DensifyStrategy(SplatfactoConfig(), num_train_data=len(out.cameras)))
    assert fio.load_checkpoint(tr2, ck) == 610
    tr2.strategy.add_mask, tr2.strategy.touch_normals = st.add_mask.clone(), st.touch_normals  # (model state, not in a ckpt)
    imp = {k: v.to(dev) for k, v in fio.import_gaussian_splat_ply(ply).items()}
    for k in PARAM_ORDER:
        assert torch.equal(tr.params[k].data, tr2.params[k].data) and torch.equal(tr.params[k].data, imp[k]), k
        sa, sb = tr.optimizers[k].state[tr.params[k]], tr2.optimizers[k].state[tr2.params[k]]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), k
    from fusionsense_amd.splatfacto import binary_opacity_active
    assert binary_opacity_active(st.cfg, tr.step)  # step 610: get_outputs thresholds the stored logits (dn_model.py:492-503)
    with torch.no_grad():
        a, b = tr.forward(out.cameras[1]), tr2.forward(out.cameras[1])
        c = render_fusionsense_fused(imp, out.cameras[1], sh_degree=tr._sh_degree_now(), device=dev,
                                     binary_threshold=st.cfg.binary_opacities_threshold, add_mask=st.add_mask)
    for k in ("rgb", "depth", "normal", "accumulation"):
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
    assert torch.equal(tr.params["opacities"].data, tr2.params["opacities"].data)
    # the reloaded trainer continues: one more step from the same state gives the same loss (to the atomics' noise)
    la, _ = tr.train_step(out.cameras[2], batches[2])
    lb, _ = tr2.train_step(out.cameras[2], batches[2])
    assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(la))


def test_monocular_depth_term_on_both_trainer_routes(dev):
    """A batch that carries "mono_depth" (dn_model.py:737-750; FusionSense's own batches do not): the EdgeAwareLogL1 term
    against the second target is evaluated by the loss node — pinned to the reference's execution in
    test_fusion_loss_kernels_match_reference_goldens[mono*] — and its depth gradient reaches the parameters the same
    way on the tape-free step (gradient images formed in the forward's pass, seed known) and on the autograd tape."""
    from fusionsense_amd.losses import LossConfig
    from fusionsense_amd.trainer import PARAM_ORDER, SplatTrainer
    W, H = 160, 96
    params, cams = _small_fusionsense(dev, W, H)
    batch = _synthetic_batch(H, W, 5)
    g = torch.Generator().manual_seed(8)
    batch["mono_depth"] = 0.3 + 1.5 * torch.rand(H, W, 1, generator=g)
    cfg = LossConfig()
    res = {}
    for name, b, direct in (("with", batch, True), ("tape", batch, False),
                            ("without", {k: v for k, v in batch.items() if k != "mono_depth"}, True)):
        tr = SplatTrainer(params, dev, sh_degree=3, loss_cfg=cfg, direct=direct)
        loss, _ = tr.train_step(cams[1], _fb(b, dev, cfg), optimizer_step=False)
        res[name] = (float(loss), {k: tr.slab.views[k].clone() for k in PARAM_ORDER})
    assert abs(res["with"][0] - res["tape"][0]) < 1e-6 * abs(res["tape"][0])
    assert res["with"][0] > res["without"][0] + 1e-3, "the term adds to the loss"
    for k in PARAM_ORDER:
        assert rel_err(res["with"][1][k], res["tape"][1][k]) < 1e-3, k
    assert rel_err(res["with"][1]["means"], res["without"][1]["means"]) > 1e-2, "and to the gradients"


@pytest.mark.parametrize("switch", ["normals_from_depth", "edge_aware_tv_cosine_logl1"])
def test_other_loss_switches_on_both_trainer_routes_and_against_the_oracle(dev, switch):
    """The get_loss_dict switches that FusionSense's CLI leaves at other values (normal_supervision = "depth" is the
    MODEL's default, dn_model.py:84; EdgeAwareTV, the cosine term and the plain depth losses are one flag away) through
    the whole step: the same loss and parameter gradients on the tape-free step and on the autograd tape, and the loss
    value equal to oracle/loss_ref (pinned by the reference-execution goldens) on the HIP render."""
    from fusionsense_amd.losses import LossConfig
    from fusionsense_amd.trainer import PARAM_ORDER, SplatTrainer
    from oracle import loss_ref
    W, H = 160, 96
    params, cams = _small_fusionsense(dev, W, H)
    batch = _synthetic_batch(H, W, 6)
    kw = (dict(normal_supervision="depth") if switch == "normals_from_depth" else
          dict(smooth_loss_type="EdgeAwareTV", use_normal_cosine_loss=True, depth_loss_type="LogL1"))
    cfg = LossConfig(**kw)
    cam = cams[1]
    res = {}
    for name, direct in (("direct", True), ("tape", False)):
        tr = SplatTrainer(params, dev, sh_degree=3, loss_cfg=cfg, direct=direct)
        loss, out = tr.train_step(cam, _fb(batch, dev, cfg), optimizer_step=False)
        res[name] = (float(loss), {k: tr.slab.views[k].clone() for k in PARAM_ORDER}, out)
    assert abs(res["direct"][0] - res["tape"][0]) < 1e-6 * abs(res["tape"][0])
    for k in PARAM_ORDER:
        assert rel_err(res["direct"][1][k], res["tape"][1][k]) < 1e-3, k
    plain = SplatTrainer(params, dev, sh_degree=3, loss_cfg=LossConfig(), direct=True)
    l0, _ = plain.train_step(cam, _fb(batch, dev, LossConfig()), optimizer_step=False)
    assert abs(float(l0) - res["direct"][0]) > 1e-4, "the switch changes the loss"
    out = res["direct"][2]
    o = {k: out[k].detach().cpu().double() for k in ("rgb", "depth", "normal")}
    b = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in batch.items()}
    ref = loss_ref.get_loss_dict(o, b, params["scales"].double(), cfg=loss_ref.LossConfig(**kw),
                                 intrinsics=(cam.fx, cam.fy, cam.cx, cam.cy))["main_loss"]
    assert abs(res["direct"][0] - float(ref)) < 1e-4 * abs(float(ref)), (res["direct"][0], float(ref))


def test_bench_line_carries_the_contract_fields():
    """One small single-GPU run of bench.py as the driver invokes it: ONE JSON line with the contract's fields — metric /
    value / unit / n_gpus / steps / warmup / ms_per_step / scaling / dtype / data / config.workload, the `roofline` object
    of the dominant kernel (bound, achieved, peak, unit, frac, traffic), `cpu_baseline` (value, unit, cores, kind, sample)
    — and this round's additions (L, pair ratio, tuner settled before the timed region)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--n-gauss", "30000",
                        "--res", "256", "--views", "4", "--no-dropin", "--cpu-crop", "32", "--cpu-views", "1",
                        "--cpu-threads", "8"], capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"] == "train_iters_per_s" and d["unit"] == "iters/s" and d["n_gpus"] == 1 and d["steps"] == 6
    assert d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=2e-3)
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0
    assert ro["frac"] == pytest.approx(ro["achieved"] / ro["peak"], rel=1e-3) and 0 < ro["frac"] < 1 and "traffic" in ro
    cb = d["cpu_baseline"]
    assert cb["kind"].startswith("port") and "extrapolated" in cb["kind"] and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb and cb["host_cores_available"] >= cb["cores"]
    c = d["config"]
    assert c["mean_walked_list"] > 1 and 0 < c["binned_over_rect_pairs"] <= 1 and c["max_walked_list"] >= c["mean_walked_list"]
    assert d["bwd_dispatch_tuning_frames_in_timed_region"] == 0 and d["device_mallocs_in_timed_region"] == 0
    assert 0 < d["iter_hbm_frac"] < 1 and d["iter_algorithmic_bytes"] <= d["iter_algorithmic_bytes_rect_pairs"]


def test_knn_kernel_is_sklearn_minus_its_first_column(dev):
    """fsgs_knn_points (what inference.knn_drop_first runs on the GPU) against sklearn's NearestNeighbors — the library
    behind the reference's knn_sk (utils/knn.py:29-44): k + 1 neighbours, first column dropped; k = 16 (knn_to_track) and
    the long-list instantiation; duplicate points keep ascending index order; a frame-sized query set runs in one launch."""
    from sklearn.neighbors import NearestNeighbors
    from fusionsense_amd.inference import knn_drop_first
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5000, 3, generator=g)
    x[100:110] = x[50]  # duplicates: ties on the distance
    y = torch.randn(1777, 3, generator=g)
    for k in (1, 16, 24):
        ref = NearestNeighbors(n_neighbors=k + 1, algorithm="brute", metric="euclidean").fit(x.double().numpy()).kneighbors(y.double().numpy())
        got = knn_drop_first(x.to(dev), y.to(dev), k).cpu()
        assert got.shape == (1777, k)
        # compare through the distances (sklearn's order among exact ties is unspecified)
        d_got = (y[:, None, :] - x[got]).double().norm(dim=-1)
        assert torch.allclose(d_got, torch.from_numpy(ref[0][:, 1:]), rtol=1e-5, atol=1e-6), k
        same = (got.numpy() == ref[1][:, 1:]).mean()
        assert same > 0.995, (k, same)
    # the tied points come out in ascending index order
    q = x[50:51].to(dev)
    idx = knn_drop_first(x.to(dev), q, 10).cpu()[0].tolist()
    tied = [i for i in idx if i == 50 or 100 <= i < 110]
    assert tied == sorted(tied) and len(tied) >= 9
    # and equals the chunked torch formulation bit for bit on the device-free path's own arithmetic
    x2 = torch.randn(5000, 3, generator=g)  # (no duplicates: torch.topk's order among exact ties is unspecified too)
    cpu = knn_drop_first(x2, y[:300], 16)
    assert torch.equal(cpu, knn_drop_first(x2.to(dev), y[:300].to(dev), 16).cpu())
    big = knn_drop_first(torch.randn(300_000, 3, generator=g).to(dev), torch.randn(200_000, 3, generator=g).to(dev), 16)
    assert big.shape == (200_000, 16) and int(big.min()) >= 0 and int(big.max()) < 300_000
