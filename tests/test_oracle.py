"""CPU tests of the oracle: analytic known answers from the published 3DGS / gsplat algorithm
(independent of any fixture), the oracle's own VJP against autograd, and the committed vectors."""
import math
import os

import numpy as np
import pytest
import torch

from fusionsense_amd import scenes
from oracle import gsplat_ref as R
from tests.helpers import activated, adversarial_gaussians, camera_mats

GOLD = os.path.join(os.path.dirname(__file__), "golden")
F = 100.0


def _one(mean, scale, opac, color, W=32, H=32, cx=16.5, cy=16.5, quat=(1, 0, 0, 0)):
    means = torch.tensor([mean], dtype=torch.float64)
    quats = torch.tensor([quat], dtype=torch.float64)
    scales = torch.tensor([scale], dtype=torch.float64)
    viewmat = torch.eye(4, dtype=torch.float64)[None]
    K = torch.tensor([[[F, 0, cx], [0, F, cy], [0, 0, 1]]], dtype=torch.float64)
    return means, quats, scales, torch.tensor([opac], dtype=torch.float64), \
        torch.tensor([color], dtype=torch.float64), viewmat, K, W, H


def test_projection_known_answer():
    """Isotropic Gaussian on the optical axis: Sigma2 = (f s / z)^2 I + 0.3 I, mean2d = (cx,cy),
    radius = ceil(3 sqrt(lambda_max)), compensation = sqrt(det0/det)."""
    s, z = 0.05, 2.0
    means, quats, scales, _, _, vm, K, W, H = _one([0, 0, z], [s, s, s], 0.5, [1, 1, 1])
    radii, m2, dp, cn, comp = R.project(means, quats, scales, vm, K, W, H)
    var = (F * s / z) ** 2
    assert torch.allclose(m2[0, 0], torch.tensor([16.5, 16.5], dtype=torch.float64))
    assert abs(dp[0, 0].item() - z) < 1e-12
    assert abs(cn[0, 0, 0].item() - 1 / (var + 0.3)) < 1e-12 and abs(cn[0, 0, 1].item()) < 1e-12
    assert radii[0, 0].item() == math.ceil(3 * math.sqrt(var + 0.3))
    assert abs(comp[0, 0].item() - var / (var + 0.3)) < 1e-12


def test_projection_culling_rules():
    params, cam = scenes.cube_scene(10)
    viewmat, K = camera_mats(cam)
    means, quats, scl, _ = adversarial_gaussians(cam)
    radii = R.project(means, quats, scl, viewmat, K, cam.width, cam.height)[0][0]
    assert radii[0] == 0 and radii[1] == 0, "behind camera / nearer than near plane"
    assert radii[2] == 0 and radii[3] == 0, "bbox fully off-image"
    assert radii[7] > 0, "image-covering Gaussian stays"
    assert radii[6] == math.ceil(3 * math.sqrt(0.3 + 1e-9)) or radii[6] == 2, "eps2d-dominated radius"


def test_raster_single_gaussian_centre_alpha():
    """alpha at the Gaussian's centre pixel = opacity; colour = c * opacity; background pixels 0."""
    means, quats, scales, op, col, vm, K, W, H = _one([0, 0, 2.0], [0.05] * 3, 0.6, [0.2, 0.4, 0.8])
    render, alpha, meta = R.rasterization(means, quats, scales, op, col, vm, K, W, H, packed=False)
    assert abs(alpha[0, 16, 16, 0].item() - 0.6) < 1e-12
    assert torch.allclose(render[0, 16, 16], torch.tensor([0.12, 0.24, 0.48], dtype=torch.float64))
    assert alpha[0, 0, 0, 0].item() == 0.0
    # one pixel to the right: alpha = o * exp(-0.5 * conic_a)
    var = (F * 0.05 / 2.0) ** 2 + 0.3
    assert abs(alpha[0, 16, 17, 0].item() - 0.6 * math.exp(-0.5 / var)) < 1e-12


def test_raster_order_saturation_early_out_and_skip():
    def render(zs, opacs, cols):
        n = len(zs)
        means = torch.tensor([[0, 0, z] for z in zs], dtype=torch.float64)
        quats = torch.tensor([[1, 0, 0, 0]] * n, dtype=torch.float64)
        scales = torch.tensor([[0.05 * z / 2.0] * 3 for z in zs], dtype=torch.float64)  # same footprint
        vm = torch.eye(4, dtype=torch.float64)[None]
        K = torch.tensor([[[F, 0, 16.5], [0, F, 16.5], [0, 0, 1]]], dtype=torch.float64)
        return R.rasterization(means, quats, scales, torch.tensor(opacs, dtype=torch.float64),
                               torch.tensor(cols, dtype=torch.float64), vm, K, 32, 32, packed=False)
    # two overlapping: nearer one first regardless of storage order
    r, a, m = render([3.0, 2.0], [0.5, 0.5], [[1, 0, 0], [0, 1, 0]])
    assert torch.allclose(r[0, 16, 16], torch.tensor([0.25, 0.5, 0.0], dtype=torch.float64))
    assert abs(a[0, 16, 16, 0].item() - 0.75) < 1e-12
    assert m["flatten_ids"][m["isect_offsets"][0, 1, 1]].item() == 1, "depth-sorted: index 1 (z=2) first"
    # saturation: opacity 1 is capped at 0.999
    r, a, m = render([2.0], [1.0], [[1, 1, 1]])
    assert abs(a[0, 16, 16, 0].item() - 0.999) < 1e-12
    # early out: T after #1 = 1e-3; #2 would give 1e-6 <= 1e-4 -> excluded, pixel stops
    r, a, m = render([2.0, 2.5, 3.0], [1.0, 1.0, 1.0], [[1, 0, 0], [0, 1, 0], [0, 0, 1]])
    assert abs(a[0, 16, 16, 0].item() - 0.999) < 1e-12
    assert torch.allclose(r[0, 16, 16], torch.tensor([0.999, 0.0, 0.0], dtype=torch.float64))
    assert m["last_ids"][0, 16, 16].item() == m["isect_offsets"][0, 1, 1].item()
    # alpha < 1/255 is skipped entirely
    r, a, m = render([2.0, 3.0], [0.0039, 0.5], [[1, 0, 0], [0, 1, 0]])
    assert abs(a[0, 16, 16, 0].item() - 0.5) < 1e-12 and r[0, 16, 16, 0].item() == 0.0


def test_sh_known_answers():
    c = torch.randn(3, 25, 3, dtype=torch.float64)
    z = torch.tensor([[0, 0, 2.0]] * 3, dtype=torch.float64)
    assert torch.allclose(R.spherical_harmonics(0, z, c), 0.2820947917738781 * c[:, 0])
    d1 = R.spherical_harmonics(1, z, c)
    assert torch.allclose(d1, 0.2820947917738781 * c[:, 0] + 0.48860251190292 * c[:, 2])
    x = torch.tensor([[3.0, 0, 0]] * 3, dtype=torch.float64)
    assert torch.allclose(R.spherical_harmonics(1, x, c), 0.2820947917738781 * c[:, 0] - 0.48860251190292 * c[:, 3])
    # bases are orthonormal on the sphere: Monte-Carlo Gram matrix ~ I/(4 pi) * 4 pi
    g = torch.Generator().manual_seed(0)
    v = torch.nn.functional.normalize(torch.randn(200000, 3, generator=g, dtype=torch.float64), dim=-1)
    B = R.sh_bases(4, v)
    gram = (B.T @ B) / v.shape[0] * 4 * math.pi
    assert (gram - torch.eye(25, dtype=torch.float64)).abs().max().item() < 0.03
    assert R.num_sh_bases(3) == 16 and R.num_sh_bases(0) == 1


def test_isect_key_format_and_rules():
    m2 = torch.tensor([[[24.0, 40.0], [32.0, 8.0]]])  # second: mean+radius = 48 = exact tile edge
    radii = torch.tensor([[10, 16]], dtype=torch.int32)
    depths = torch.tensor([[1.5, 0.25]])
    tpg, ids, flat = R.isect_tiles(m2, radii, depths, 16, 8, 8)
    # first: x in [14,34) -> tiles 0..2, y in [30,50) -> tiles 1..3
    assert tpg[0, 0] == 9 and tpg[0, 1] == 2 * 2  # x tiles {1,2}, y tiles {0,1}
    tb = R.tile_bits(64)
    assert tb == 7
    k0 = ids[0]
    assert (k0 >> 32) == 1 * 8 + 0 and (k0 & 0xFFFFFFFF) == np.float32(1.5).view(np.int32)
    # legacy rule adds one tile column/row when mean+radius lands exactly on a tile edge
    tpg_l, _, _ = R.isect_tiles(m2, radii, depths, 16, 8, 8, legacy=True)
    assert tpg_l[0, 1] == 3 * 2 and tpg_l[0, 0] == 9  # x tiles {1,2,3}: the extra column
    # stable sort: equal keys keep emission order
    keys = np.array([5, 3, 5, 3, 5], dtype=np.int64)
    vals = np.arange(5, dtype=np.int32)
    ks, vs = R.sort_isects(keys, vals)
    assert ks.tolist() == [3, 3, 5, 5, 5] and vs.tolist() == [1, 3, 0, 2, 4]
    offs = R.isect_offset_encode(np.array([(2 << 32) | 7, (2 << 32) | 9, (5 << 32) | 1], dtype=np.int64), 1, 4, 2)
    assert offs.reshape(-1).tolist() == [0, 0, 0, 2, 2, 2, 3, 3]


def test_viewmat_and_quat():
    c2w = scenes.look_at_c2w(torch.tensor([1.0, 2.0, 3.0]), torch.zeros(3))
    vm = R.get_viewmat(c2w[None])[0]
    p = vm @ torch.tensor([0.0, 0.0, 0.0, 1.0])
    assert abs(p[2].item() - math.sqrt(14.0)) < 1e-5 and abs(p[0].item()) < 1e-5, "target on +z (OpenCV)"
    q = torch.tensor([[0.0, 0.0, 0.0, 2.0]])  # 180 deg about z, unnormalised
    assert torch.allclose(R.quat_to_rotmat(q)[0], torch.diag(torch.tensor([-1.0, -1.0, 1.0])), atol=1e-6)


def test_oracle_manual_bwd_matches_autograd_fp64():
    params, cam = scenes.cube_scene(120, seed=5)
    act = {k: v.double() for k, v in activated(params).items()}
    viewmat, K = camera_mats(cam)
    W = H = 48
    K = K.double().clone()
    K[0, 0, 0] = K[0, 1, 1] = 48.0
    K[0, 0, 2] = K[0, 1, 2] = 24.0
    radii, m2, dp, cn, _ = R.project(act["means"], act["quats"], act["scales"], viewmat.double(), K, W, H)
    tpg, ids, flat = R.isect_tiles(m2, radii, dp, 16, 3, 3)
    ids_s, flat_s = R.sort_isects(ids, flat)
    offs = R.isect_offset_encode(ids_s, 1, 3, 3)
    g = torch.Generator().manual_seed(0)
    cols = torch.rand(1, 120, 4, generator=g, dtype=torch.float64)
    opac = act["opacities"][None].clone()
    bg = torch.rand(1, 4, generator=g, dtype=torch.float64)
    leaves = [t.clone().requires_grad_(True) for t in (m2, cn, cols, opac)]
    out, alpha, _ = R.rasterize_to_pixels(*leaves, W, H, 16, offs, flat_s, bg)
    v_r = torch.randn(out.shape, generator=g, dtype=torch.float64)
    v_a = torch.randn(alpha.shape, generator=g, dtype=torch.float64)
    torch.autograd.backward([out, alpha], [v_r, v_a])
    man = R.rasterize_to_pixels_bwd(m2, cn, cols, opac, W, H, 16, offs, flat_s, v_r, v_a, bg)
    for k, leaf in zip(("v_means2d", "v_conics", "v_colors", "v_opacities"), leaves):
        assert (man[k] - leaf.grad).abs().max().item() < 1e-9 * (1 + leaf.grad.abs().max().item()), k
    assert (man["v_means2d_abs"] >= man["v_means2d"].abs() - 1e-12).all()


def test_oracle_matches_committed_vectors():
    g = np.load(os.path.join(GOLD, "oracle_vectors.npz"))
    params, cam = scenes.cube_scene(1000, seed=0)
    viewmat, K = camera_mats(cam)
    W, H = cam.width, cam.height
    means, quats, scl, _ = adversarial_gaussians(cam)
    r = R.project(means.double(), quats.double(), scl.double(), viewmat.double(), K.double(), W, H)
    assert np.array_equal(r[0].numpy(), g["proj_radii"])
    assert np.allclose(r[1].numpy(), g["proj_means2d"], rtol=1e-12, atol=1e-12)
    assert np.allclose(r[3].numpy(), g["proj_conics"], rtol=1e-12, atol=1e-12)
    r32 = R.project(means, quats, scl, viewmat, K, W, H)
    for legacy, tag in ((False, "new"), (True, "legacy")):
        tpg, ids, flat = R.isect_tiles(r32[1], r32[0], r32[2], 16, 8, 8, legacy=legacy)
        ids_s, flat_s = R.sort_isects(ids, flat)
        assert np.array_equal(tpg, g[f"isect_{tag}_tpg"])
        assert np.array_equal(ids_s, g[f"isect_{tag}_ids_sorted"])
        assert np.array_equal(flat_s, g[f"isect_{tag}_flat_sorted"])
        assert np.array_equal(R.isect_offset_encode(ids_s, 1, 8, 8), g[f"isect_{tag}_offsets"])
    dirs = torch.nn.functional.normalize(torch.from_numpy(g["sh_dirs"]), dim=-1)
    for d in range(5):
        assert np.allclose(R.sh_bases(d, dirs).numpy(), g[f"sh_bases_deg{d}"], atol=1e-14)


def test_config1_full_frame_matches_committed_checksum():
    g = np.load(os.path.join(GOLD, "oracle_vectors.npz"))
    params, cam = scenes.cube_scene(1000, seed=0)
    viewmat, K = camera_mats(cam)
    act = {k: v.double() for k, v in activated(params).items()}
    render, alpha, meta = R.rasterization(act["means"], act["quats"], act["scales"], act["opacities"],
                                          act["colors"], viewmat.double(), K.double(), cam.width, cam.height,
                                          sh_degree=3, packed=False, render_mode="RGB+ED")
    assert np.allclose(render.mean(dim=(0, 1, 2)).numpy(), g["cfg1_render_mean"], atol=1e-12)
    assert np.allclose(render[0, 56:72, 56:72].numpy(), g["cfg1_patch"], atol=1e-12)
    assert meta["flatten_ids"].numel() == int(g["cfg1_n_isects"])
    assert int(meta["flatten_ids"].long().sum()) == int(g["cfg1_flat_sum"])
    assert np.array_equal(meta["isect_offsets"].numpy(), g["cfg1_offsets"])


LOSS_CASES = ["full", "nomask", "dense", "mono", "monoonly",
              "ndepth", "ndepth_nomask", "eatv", "cosine", "cosine_depth", "l1", "logl1", "mse",  # (round 4: the other switches)
              "huber", "sparse", "sparse_off"]  # (terms that stay torch code on top of the HIP node)


def loss_case_config(d, case, cls):
    """The LossConfig of a reference_loss.npz case (``<case>.switches`` = normal supervision, smooth type, cosine, depth
    type; absent for FusionSense's own configuration) and the camera intrinsics the depth-normal supervision read."""
    kw = {}
    if f"{case}.switches" in d.files:
        ns, sm, cs, dt = (int(x) for x in d[f"{case}.switches"])  # numeric codes, see make_reference_loss_goldens.py
        kw = dict(normal_supervision=["mono", "depth"][ns], smooth_loss_type=["TV", "EdgeAwareTV"][sm],
                  use_normal_cosine_loss=bool(cs), depth_loss_type=["EdgeAwareLogL1", "L1", "LogL1", "MSE", "HuberL1"][dt])
    intr = tuple(float(x) for x in d[f"{case}.intr"]) if f"{case}.intr" in d.files else None
    return cls(**kw), intr


@pytest.mark.parametrize("case", LOSS_CASES)
def test_loss_oracle_matches_reference_goldens(case):
    """oracle/loss_ref.dn_terms against numbers produced by EXECUTING the reference's own
    DNSplatterModel.get_loss_dict (tests/golden/make_reference_loss_goldens.py): value and the gradients with
    respect to the rendered depth / normal images and the log-scales."""
    from oracle import loss_ref
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_loss.npz"))
    t = lambda k: torch.from_numpy(d[f"{case}.{k}"])  # noqa: E731
    depth = t("depth").clone().requires_grad_(True)
    normal = t("normal").clone().requires_grad_(True)
    scales = t("scales").clone().requires_grad_(True)
    batch = {"image": t("image"), "sensor_depth": t("sensor_depth"), "normal": t("normal_gt")}
    if f"{case}.mono_depth" in d.files:  # the monocular-depth branch (dn_model.py:737-750)
        batch["mono_depth"] = t("mono_depth")
        if bool(d[f"{case}.mono_only"]):
            del batch["sensor_depth"]
    if bool(d[f"{case}.has_mask"]):
        batch["mask"] = t("mask")
    n_touch = int(d[f"{case}.n_touch"])
    add_mask = None
    if n_touch:
        add_mask = torch.zeros(scales.shape[0], dtype=torch.bool)
        add_mask[-n_touch:] = True
    out = {"rgb": t("rgb"), "depth": depth, "normal": normal}
    cfg, intr = loss_case_config(d, case, loss_ref.LossConfig)
    loss = loss_ref.dn_terms(out, batch, scales, t("normals_world"), add_mask, t("touch_normals") if n_touch else None,
                             cfg, intr)
    opac = None
    if f"{case}.sparse" in d.files:  # the sparse-opacity term (dn_model.py:821-837)
        opac = t("opacities").clone().requires_grad_(True)
        step, every, reset, refine = (int(x) for x in d[f"{case}.sparse"])
        loss = loss + loss_ref.sparse_opacity_term(opac, t("vis_indices"), step, every, reset, refine,
                                                   float(d[f"{case}.sparse_lambda"]))
    loss.backward()
    assert abs(float(loss) - float(d[f"{case}.loss"])) <= 1e-6 * abs(float(d[f"{case}.loss"]))
    for name, g in (("v_depth", depth.grad), ("v_normal", normal.grad), ("v_scales", scales.grad)):
        ref = t(name)
        assert torch.allclose(g, ref, rtol=1e-5, atol=1e-9), (name, float((g - ref).abs().max()))
    if opac is not None:
        g = opac.grad if opac.grad is not None else torch.zeros_like(opac)
        assert torch.allclose(g, t("v_opacities"), rtol=1e-5, atol=1e-10)
        assert (float(g.abs().max()) > 0) == (case == "sparse")  # ("sparse_off": inside the window after a reset)


def test_ssim_oracle_known_answers():
    """The photometric term's SSIM is recalled (torchmetrics absent): identical images -> 1, symmetric, and the
    closed form for two constant images ((2ab + c1) / (a^2 + b^2 + c1))."""
    from oracle import loss_ref
    g = torch.Generator().manual_seed(0)
    x, y = torch.rand(3, 40, 36, generator=g, dtype=torch.float64), torch.rand(3, 40, 36, generator=g, dtype=torch.float64)
    assert abs(float(loss_ref.ssim_torchmetrics(x, x)) - 1.0) < 1e-12
    assert abs(float(loss_ref.ssim_torchmetrics(x, y)) - float(loss_ref.ssim_torchmetrics(y, x))) < 1e-12
    a, b = 0.3, 0.7
    v = float(loss_ref.ssim_torchmetrics(torch.full((3, 30, 30), a, dtype=torch.float64), torch.full((3, 30, 30), b, dtype=torch.float64)))
    assert abs(v - (2 * a * b + 1e-4) / (a * a + b * b + 1e-4)) < 1e-9
    # the product's CPU mirror (separable formulation) agrees with this one
    from fusionsense_amd import losses
    assert abs(float(losses.ssim(x.float(), y.float())) - float(loss_ref.ssim_torchmetrics(x, y))) < 1e-5


# ---- goldens made by EXECUTING the reference's own DNSplatterModel methods (tests/golden/make_reference_model_goldens.py) ----
def _model_goldens():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_model.npz"))


NAMES6 = ["means", "scales", "quats", "features_dc", "features_rest", "opacities"]


@pytest.mark.parametrize("case", ["plain", "binary_anchor"])
def test_caller_oracle_matches_reference_get_outputs(case):
    """oracle/fusion_ref.render_fusionsense (the restated caller) against the reference's own get_outputs executed
    over the same rasterizer stand-in: images, the binary-opacity write, per-Gaussian normals, and the parameter
    gradients including the touch-anchor detach."""
    from fusionsense_amd.scenes import Camera
    from oracle import fusion_ref
    d = _model_goldens()
    t = lambda k: torch.from_numpy(d[k])  # noqa: E731
    fx, fy, cx, cy, W, H = [float(v) for v in d["go.intr"]]
    cam = Camera(t("go.c2w"), fx, fy, cx, cy, int(W), int(H))
    params = {k: t(f"go.in.{k}").clone().requires_grad_(True) for k in NAMES6}
    step = int(d[f"go.{case}.step"])
    add_mask = t(f"go.{case}.add_mask") if f"go.{case}.add_mask" in d.files else None
    bthr = 0.9 if fusion_ref.binary_opacity_step(step) else None
    assert (bthr is not None) == (case == "binary_anchor")
    out = fusion_ref.render_fusionsense(params, cam, sh_degree=int(d[f"go.{case}.sh_degree_to_use"]), add_mask=add_mask,
                                        binary_threshold=bthr)
    loss = sum((out[k] * t(f"go.w.{k}")).mean() for k in ("rgb", "depth", "normal"))
    loss.backward()
    for k in ("rgb", "depth", "normal", "accumulation"):
        assert torch.allclose(out[k].detach(), t(f"go.{case}.out.{k}"), atol=1e-6), k
    assert torch.equal(params["opacities"].detach(), t(f"go.{case}.opacities_after"))
    assert torch.allclose(out["normals_world"].detach(), t(f"go.{case}.normals_world"), atol=1e-6)
    for k in NAMES6:
        g = params[k].grad if params[k].grad is not None else torch.zeros_like(params[k])
        ref = t(f"go.{case}.grad.{k}")
        assert torch.allclose(g, ref, rtol=1e-4, atol=1e-8 + 1e-5 * float(ref.abs().max())), (k, float((g - ref).abs().max()))
    if add_mask is not None:
        assert float(t(f"go.{case}.grad.means")[add_mask].abs().max()) == 0.0  # (what the golden itself says)


@pytest.mark.parametrize("step", [400, 700, 3100, 3500, 10000])
def test_densify_oracle_matches_reference_refinement_after(step):
    """oracle/splatfacto_ref.refinement_after against the reference's own refinement_after (its schedule, masks,
    the re-computed dups after the in-place shrink, cull and opacity reset), executed over stand-ins for the inherited
    nerfstudio methods: parameters and both Adam moments, row for row."""
    from fusionsense_amd.splatfacto import SplatfactoConfig
    from oracle import splatfacto_ref
    d = _model_goldens()
    t = lambda k: torch.from_numpy(d[k])  # noqa: E731
    names = NAMES6 + ["normals"]
    pre = {k: t(f"ra.{step}.pre.{k}") for k in names}
    adam = {k: {"exp_avg": t(f"ra.{step}.pre_m.{k}"), "exp_avg_sq": t(f"ra.{step}.pre_v.{k}")} for k in names}
    state = {k: t(f"ra.stats.{k}").clone() for k in ("xys_grad_norm", "vis_counts", "max_2Dsize")}
    add_mask = None
    if step == 3500:
        add_mask = torch.zeros(pre["means"].shape[0], dtype=torch.bool)
        add_mask[:50] = True
    post, post_adam, rep = splatfacto_ref.refinement_after(pre, adam, state, SplatfactoConfig(), step, 9, (720, 1280),
                                                           t(f"ra.{step}.randn"), add_mask)
    for k in names:
        assert post[k].shape == t(f"ra.{step}.post.{k}").shape, (k, post[k].shape, rep)
        assert torch.allclose(post[k], t(f"ra.{step}.post.{k}"), atol=1e-6), k
        assert torch.equal(post_adam[k]["exp_avg"], t(f"ra.{step}.post_m.{k}")), k
        assert torch.equal(post_adam[k]["exp_avg_sq"], t(f"ra.{step}.post_v.{k}")), k
    assert bool(d[f"ra.{step}.stats_cleared"]) == (step > 500)
