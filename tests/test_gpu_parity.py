"""GPU parity: the HIP path (through the C-ABI, via fusionsense_amd.ops) against the CPU oracle
on the same seeded inputs.  Bit-exact for integer outputs, stated tolerances for fp32."""
import math

import numpy as np
import pytest
import torch

from fusionsense_amd import scenes
from oracle import gsplat_ref as R
from tests.helpers import activated, adversarial_gaussians, camera_mats, oracle_frame, poison_capacity, rel_err

pytestmark = pytest.mark.gpu

FWD_RTOL = 1e-4   # fp32 forward tolerance (relative to the tensor's max magnitude)
BWD_RTOL = 2e-3   # fp32 backward tolerance (atomics reorder the sums)
E2E_RTOL = 3e-3   # end-to-end parameter gradients, per tensor relative to that tensor's own maximum


def test_library_loads_on_gpu(dev):
    from fusionsense_amd import _lib
    assert _lib.load().fsgs_version() >= 100


@pytest.mark.parametrize("scene", ["cube", "adversarial"])
def test_project_fwd(dev, scene):
    from fusionsense_amd import ops
    params, cam = scenes.cube_scene(1000, seed=0)
    viewmat, K = camera_mats(cam)
    if scene == "cube":
        act = activated(params)
        means, quats, scl = act["means"], act["quats"], act["scales"]
    else:
        means, quats, scl, _ = adversarial_gaussians(cam)
    W, H = cam.width, cam.height
    r64 = R.project(means.double(), quats.double(), scl.double(), viewmat.double(), K.double(), W, H)
    r32 = R.project(means, quats, scl, viewmat, K, W, H)
    radii, m2, dp, cn, comp = ops.project_fwd(means.to(dev), quats.to(dev).contiguous(), scl.to(dev),
                                              viewmat.to(dev), K.to(dev), W, H, 0.3, 0.01, 1e10, 0.0, True)
    radii_c = radii.cpu()
    # radii: exact, except where the fp64 pre-ceil value sits within 1e-3 of an integer or a cull
    # decision is borderline (fp32 vs fp64 oracles disagree)
    stable = r64[0] == r32[0]
    mism = (radii_c != r32[0]) & stable
    assert mism.sum().item() == 0, f"{mism.sum().item()} radii differ: {radii_c[mism][:8]} vs {r32[0][mism][:8]}"
    vis = (radii_c > 0) & (r32[0] > 0)
    assert vis.sum() > 0
    for name, got, ref in (("means2d", m2, r32[1]), ("depths", dp, r32[2]), ("conics", cn, r32[3]),
                           ("comp", comp, r32[4])):
        g = got.cpu()[vis]
        r = ref[vis]
        err = ((g - r).abs() / (r.abs() + 1.0)).max().item()
        assert err < FWD_RTOL, f"{name}: {err}"
    # culled Gaussians are zeroed
    assert (m2.cpu()[radii_c == 0] == 0).all() and (cn.cpu()[radii_c == 0] == 0).all()


@pytest.mark.parametrize("legacy", [False, True])
def test_isect_sort_offsets_bit_exact(dev, legacy):
    from fusionsense_amd import ops
    params, cam = scenes.cube_scene(3000, seed=3)
    act = activated(params)
    viewmat, K = camera_mats(cam)
    W, H = cam.width, cam.height
    radii, m2, dp, cn, _ = R.project(act["means"], act["quats"], act["scales"], viewmat, K, W, H)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids, flat = R.isect_tiles(m2, radii, dp, 16, tw, th, legacy=legacy)
    ids_s, flat_s = R.sort_isects(ids, flat)
    offs = R.isect_offset_encode(ids_s, 1, tw, th)
    g_tpg, g_ids, g_flat = ops.isect_tiles(m2.to(dev), radii.to(dev), dp.to(dev), 16, tw, th, legacy=legacy,
                                           sort=False)
    assert np.array_equal(g_tpg.cpu().numpy(), tpg)
    assert np.array_equal(g_ids.cpu().numpy(), ids)
    assert np.array_equal(g_flat.cpu().numpy(), flat)
    g_tpg, g_ids, g_flat = ops.isect_tiles(m2.to(dev), radii.to(dev), dp.to(dev), 16, tw, th, legacy=legacy,
                                           sort=True)
    assert np.array_equal(g_ids.cpu().numpy(), ids_s)
    assert np.array_equal(g_flat.cpu().numpy(), flat_s)
    g_off = ops.isect_offset_encode(g_ids, 1, tw, th)
    assert np.array_equal(g_off.cpu().numpy(), offs)


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 2047, 2048, 2049, 100_003, 1_500_000, 5_000_001])
def test_sort_pairs_stable(dev, n):
    from fusionsense_amd import ops
    rng = np.random.default_rng(n)
    # few distinct keys -> many ties: stability is visible in the payload order
    keys = rng.integers(0, 1 << 44, size=n, dtype=np.int64)
    if n > 10:
        keys[rng.integers(0, n, size=n // 2)] = keys[0]
    vals = np.arange(n, dtype=np.int32)
    order = np.argsort(keys.view(np.uint64), kind="stable")
    k, v = ops.sort_pairs(torch.from_numpy(keys).to(dev), torch.from_numpy(vals).to(dev), end_bit=44)
    assert np.array_equal(k.cpu().numpy(), keys[order])
    assert np.array_equal(v.cpu().numpy(), vals[order])


def test_sort_pairs_full_64bit(dev):
    from fusionsense_amd import ops
    rng = np.random.default_rng(7)
    keys = rng.integers(0, np.iinfo(np.int64).max, size=300_000, dtype=np.int64)
    vals = rng.integers(0, 1 << 31, size=keys.size, dtype=np.int32)
    order = np.argsort(keys.view(np.uint64), kind="stable")
    k, v = ops.sort_pairs(torch.from_numpy(keys).to(dev), torch.from_numpy(vals).to(dev), end_bit=64)
    assert np.array_equal(k.cpu().numpy(), keys[order])
    assert np.array_equal(v.cpu().numpy(), vals[order])


def _frame(seed=0, n=1000, with_bg=False, D=4):
    params, cam = scenes.cube_scene(n, seed=seed)
    act = activated(params)
    viewmat, K = camera_mats(cam)
    W, H = cam.width, cam.height
    fr = oracle_frame(act, viewmat, K, W, H)
    g = torch.Generator().manual_seed(seed + 99)
    cols = torch.rand(1, n, D, generator=g)
    opac = act["opacities"][None].contiguous()
    bg = torch.rand(1, D, generator=g) if with_bg else None
    return cam, fr, cols, opac, bg


@pytest.mark.parametrize("live", [False, True])
@pytest.mark.parametrize("D,with_bg", [(4, False), (3, True), (1, False)])
def test_raster_fwd(dev, D, with_bg, live):
    from fusionsense_amd import ops
    cam, fr, cols, opac, bg = _frame(seed=1, D=D, with_bg=with_bg)
    ids = torch.from_numpy(fr["ids_s"]).to(dev) if live else None
    W, H = cam.width, cam.height
    ref, ref_a, ref_last = R.rasterize_to_pixels(fr["means2d"], fr["conics"], cols, opac, W, H, 16,
                                                 fr["offsets"], fr["flat_s"], bg)
    out, alpha, last = ops._Rasterize.apply(
        fr["means2d"].to(dev), fr["conics"].to(dev), cols.to(dev), opac.to(dev),
        bg.to(dev) if bg is not None else None, torch.from_numpy(fr["offsets"]).to(dev),
        torch.from_numpy(fr["flat_s"]).to(dev), W, H, 16, False, ids)
    assert (out.cpu() - ref).abs().max().item() < FWD_RTOL * max(1.0, ref.abs().max().item())
    assert (alpha.cpu() - ref_a).abs().max().item() < FWD_RTOL
    # last_ids: bit-exact wherever the threshold decisions (alpha < 1/255, T <= 1e-4) are not fp32-borderline, i.e.
    # wherever the oracle in fp32 and in fp64 arrive at the same index
    _, _, ref_last64 = R.rasterize_to_pixels(fr["means2d"].double(), fr["conics"].double(), cols.double(), opac.double(),
                                             W, H, 16, fr["offsets"], fr["flat_s"], bg.double() if bg is not None else None)
    stable = ref_last == ref_last64
    assert stable.mean() > 0.999, "the oracle itself must be decided on (almost) every pixel"
    assert np.array_equal(last.cpu().numpy()[stable], ref_last[stable]), "last_ids must be exact where the oracle is decided"


@pytest.mark.parametrize("live", [False, True])
@pytest.mark.parametrize("D,with_bg", [(4, False), (3, True)])
def test_raster_bwd(dev, D, with_bg, live):
    from fusionsense_amd import ops
    cam, fr, cols, opac, bg = _frame(seed=2, D=D, with_bg=with_bg)
    ids = torch.from_numpy(fr["ids_s"]).to(dev) if live else None
    W, H = cam.width, cam.height
    g = torch.Generator().manual_seed(5)
    v_render = torch.randn(1, H, W, D, generator=g)
    v_alpha = torch.randn(1, H, W, 1, generator=g)
    ref = R.rasterize_to_pixels_bwd(fr["means2d"], fr["conics"], cols, opac, W, H, 16, fr["offsets"],
                                    fr["flat_s"], v_render, v_alpha, bg)
    ins = [t.to(dev).requires_grad_(True) for t in (fr["means2d"], fr["conics"], cols, opac)]
    out, alpha, _ = ops._Rasterize.apply(*ins, bg.to(dev) if bg is not None else None,
                                         torch.from_numpy(fr["offsets"]).to(dev),
                                         torch.from_numpy(fr["flat_s"]).to(dev), W, H, 16, True, ids)
    torch.autograd.backward([out, alpha], [v_render.to(dev), v_alpha.to(dev)])
    got = dict(v_means2d=ins[0].grad, v_conics=ins[1].grad, v_colors=ins[2].grad, v_opacities=ins[3].grad,
               v_means2d_abs=ins[0].absgrad)
    for k, v in got.items():
        e = rel_err(v, ref[k])
        assert e < BWD_RTOL, f"{k}: rel err {e}"


@pytest.mark.parametrize("degree", [0, 1, 2, 3, 4])
def test_sh_fwd_bwd(dev, degree):
    from fusionsense_amd import ops
    g = torch.Generator().manual_seed(degree)
    N, K, C = 777, 25 if degree == 4 else 16, 2
    means = torch.randn(N, 3, generator=g)
    campos = torch.randn(C, 3, generator=g)
    coeffs = torch.randn(N, K, 3, generator=g) * 0.5
    radii = (torch.rand(C, N, generator=g) > 0.2).to(torch.int32) * 3
    depths = torch.rand(C, N, generator=g)
    v = torch.randn(C, N, 4, generator=g)

    m_r = means.clone().requires_grad_(True)
    c_r = coeffs.clone().requires_grad_(True)
    d_r = depths.clone().requires_grad_(True)
    dirs = m_r[None] - campos[:, None]
    col = R.spherical_harmonics(degree, dirs, c_r[None].expand(C, -1, -1, -1), masks=radii > 0)
    col = torch.cat([torch.clamp_min(col + 0.5, 0.0), d_r[..., None]], -1)
    col.backward(v)

    m_g = means.to(dev).requires_grad_(True)
    c_g = coeffs.to(dev).requires_grad_(True)
    d_g = depths.to(dev).requires_grad_(True)
    out = ops._SHColors.apply(m_g, c_g, campos.to(dev), radii.to(dev), d_g, degree)
    out.backward(v.to(dev))
    assert (out.cpu() - col.detach()).abs().max().item() < 1e-5
    assert rel_err(c_g.grad, c_r.grad) < 1e-5
    m_ref = m_r.grad if m_r.grad is not None else torch.zeros_like(means)  # degree 0: no dependence
    assert (m_g.grad.cpu() - m_ref).abs().max().item() < 1e-4 * (1 + m_ref.abs().max().item())
    assert rel_err(d_g.grad, d_r.grad) < 1e-6


@pytest.mark.parametrize("degree", [0, 2, 3])
def test_sh_gradient_from_factors(dev, degree):
    """The data-parallel exchange format: fsgs_sh_bwd_colors (masked colour gradient + camera centre, [N + 1,4]) +
    fsgs_sh_coeff_grad over R gathered views against fsgs_sh_bwd_split per view: R = 1 reproduces the coefficient
    gradients and v_means bit for bit; R = 2 equals the mean of the two views' gradients (and the oracle's)."""
    from fusionsense_amd._lib import load, ptr, stream_ptr
    lib = load()
    g = torch.Generator().manual_seed(20 + degree)
    N, K = 1003, 16
    means = torch.randn(N, 3, generator=g).to(dev)
    dc = (torch.randn(N, 3, generator=g) * 0.5).to(dev)
    rest = (torch.randn(N, K - 1, 3, generator=g) * 0.5).to(dev)
    sp = stream_ptr(dev)
    views = []
    for r in range(2):
        campos = torch.randn(3, generator=g).to(dev)
        radii = ((torch.rand(1, N, generator=g) > 0.25).to(torch.int32) * 4).to(dev)
        v_packed = torch.randn(N, 16, generator=g).to(dev)
        # per-view reference: the ordinary backward
        g_dc, g_rest, g_means = torch.empty_like(dc), torch.empty_like(rest), torch.empty_like(means)
        assert lib.fsgs_sh_bwd_split(1, N, K, degree, ptr(means), ptr(campos), ptr(dc), ptr(rest), ptr(radii), 16,
                                     ptr(v_packed), ptr(g_dc), ptr(g_rest), ptr(g_means), None, 1, 0, sp) == 0
        fac = torch.full((N + 1, 4), float("nan"), device=dev)
        g_means2 = torch.empty_like(means)
        assert lib.fsgs_sh_bwd_colors(N, K, degree, ptr(means), ptr(campos), ptr(dc), ptr(rest), ptr(radii), 16,
                                      ptr(v_packed), ptr(fac), ptr(g_means2), 1, 0, sp) == 0
        assert torch.equal(g_means2, g_means)
        assert torch.equal(fac[N, :3], campos) and bool((fac[:N, :3][radii[0] <= 0] == 0).all())
        r_dc, r_rest = torch.empty_like(dc), torch.empty_like(rest)
        assert lib.fsgs_sh_coeff_grad(1, N, K, degree, ptr(means), ptr(fac), 1.0, ptr(r_dc), ptr(r_rest), sp) == 0
        assert torch.equal(r_dc, g_dc) and torch.equal(r_rest, g_rest)
        views.append((fac, g_dc, g_rest))
    gathered = torch.stack([views[0][0], views[1][0]]).contiguous()
    m_dc, m_rest = torch.empty_like(dc), torch.empty_like(rest)
    assert lib.fsgs_sh_coeff_grad(2, N, K, degree, ptr(means), ptr(gathered), 0.5, ptr(m_dc), ptr(m_rest), sp) == 0
    want_dc, want_rest = 0.5 * (views[0][1] + views[1][1]), 0.5 * (views[0][2] + views[1][2])
    assert (m_dc - want_dc).abs().max().item() <= 1e-6 * (1 + want_dc.abs().max().item())
    assert (m_rest - want_rest).abs().max().item() <= 1e-6 * (1 + want_rest.abs().max().item())
    if degree < 3:
        kk = (degree + 1) ** 2
        assert bool((m_rest[:, kk - 1:] == 0).all()), "bands above sh_degree_to_use get no gradient"
    # ... and applied in the same launch (fsgs_sh_coeff_grad_adam) = fsgs_sh_coeff_grad + fsgs_adam_step
    from fusionsense_amd import ops
    p1 = [dc.clone(), rest.clone()]
    m1 = [torch.rand_like(dc) * 1e-3, torch.rand_like(rest) * 1e-3]
    v1 = [torch.rand_like(dc) * 1e-6, torch.rand_like(rest) * 1e-6]
    p2, m2, v2 = [t.clone() for t in p1], [t.clone() for t in m1], [t.clone() for t in v1]
    lrs, step = (0.0025, 0.000125), 7
    ops.adam_step_(p1, [m_dc, m_rest], m1, v1, lrs, step, 0.9, 0.999, 1e-15)
    assert lib.fsgs_sh_coeff_grad_adam(2, N, K, degree, ptr(means), ptr(gathered), 0.5, ptr(p2[0]), ptr(m2[0]),
                                       ptr(v2[0]), lrs[0], ptr(p2[1]), ptr(m2[1]), ptr(v2[1]), lrs[1], step, 0.9, 0.999,
                                       1e-15, sp) == 0
    for x, y in zip(p1 + m1 + v1, p2 + m2 + v2):
        assert (x - y).abs().max().item() <= 1e-6 * (1e-3 + x.abs().max().item())


@pytest.mark.parametrize("scene,antialiased", [("cube", False), ("adversarial", False), ("cube", True)])
def test_project_bwd(dev, scene, antialiased):
    from fusionsense_amd import ops
    params, cam = scenes.cube_scene(500, seed=4)
    viewmat, K = camera_mats(cam)
    if scene == "cube":
        act = activated(params)
        means, quats, scl = act["means"], act["quats"] * 1.7, act["scales"]
    else:
        means, quats, scl, _ = adversarial_gaussians(cam)
    W, H = cam.width, cam.height
    g = torch.Generator().manual_seed(11)
    N = means.shape[0]
    v_m2, v_dp, v_cn, v_cp = (torch.randn(1, N, 2, generator=g), torch.randn(1, N, generator=g),
                              torch.randn(1, N, 3, generator=g) * 1e-2, torch.randn(1, N, generator=g))
    leaves = [t.clone().double().requires_grad_(True) for t in (means, quats, scl)]
    vm64 = viewmat.double().requires_grad_(True)
    radii, m2, dp, cn, cp = R.project(*leaves, vm64, K.double(), W, H)
    loss = (m2 * v_m2).sum() + (dp * v_dp).sum() + (cn * v_cn).sum()
    if antialiased:
        loss = loss + (cp * v_cp).sum()
    loss.backward()

    gl = [t.to(dev).contiguous().requires_grad_(True) for t in (means, quats, scl)]
    vm = viewmat.to(dev).requires_grad_(True)
    r_g, m2_g, dp_g, cn_g, cp_g = ops._Projection.apply(*gl, vm, K.to(dev), W, H, 0.3, 0.01, 1e10, 0.0,
                                                        antialiased)
    same = (r_g.cpu() == radii)
    loss_g = (m2_g * v_m2.to(dev)).sum() + (dp_g * v_dp.to(dev)).sum() + (cn_g * v_cn.to(dev)).sum()
    if antialiased:
        loss_g = loss_g + (cp_g * v_cp.to(dev)).sum()
    loss_g.backward()
    assert same.all()
    for name, a, b in (("means", gl[0].grad, leaves[0].grad), ("quats", gl[1].grad, leaves[1].grad),
                       ("scales", gl[2].grad, leaves[2].grad), ("viewmats", vm.grad, vm64.grad)):
        assert rel_err(a, b) < 5e-4, f"{name}: {rel_err(a, b)}"


def test_normals_fwd_bwd(dev):
    from fusionsense_amd import ops
    params, cam = scenes.cube_scene(2000, seed=8)
    q = (params["quats"] * 2.5).clone()
    ls = params["scales"].clone()
    ls[:10] = ls[:10, :1]  # ties in argmin -> first index
    v = torch.randn(2000, 3, generator=torch.Generator().manual_seed(1))
    q_r = q.clone().double().requires_grad_(True)
    nw, nc = R.gaussian_normals(q_r, ls.double(), params["means"].double(), cam.c2w.double())
    (nc * v).sum().backward()
    q_g = q.to(dev).requires_grad_(True)
    nw_g, nc_g = ops.gaussian_normals(q_g, ls.to(dev), params["means"].to(dev), cam.c2w.to(dev))
    (nc_g * v.to(dev)).sum().backward()
    assert (nw_g.cpu() - nw.detach().float()).abs().max().item() < 1e-5
    assert (nc_g.cpu() - nc.detach().float()).abs().max().item() < 1e-5
    assert rel_err(q_g.grad, q_r.grad) < 1e-4


def test_end_to_end_get_outputs(dev):
    """render_fusionsense (= DNSplatterModel.get_outputs) forward + backward on config #1."""
    from fusionsense_amd.fusion import render_fusionsense
    from oracle.fusion_ref import render_fusionsense as render_ref
    params, cam = scenes.cube_scene(1000, seed=0)
    gp = {k: v.to(dev).requires_grad_(True) for k, v in params.items()}
    out = render_fusionsense(gp, cam, sh_degree=3, device=dev)
    g = torch.Generator().manual_seed(3)
    w_rgb, w_d, w_n = (torch.rand(128, 128, 3, generator=g), torch.rand(128, 128, 1, generator=g),
                       torch.rand(128, 128, 3, generator=g))
    loss = (out["rgb"] * w_rgb.to(dev)).mean() + (out["depth"] * w_d.to(dev)).mean() + \
        (out["normal"] * w_n.to(dev)).mean()
    loss.backward()
    cp = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = render_ref(cp, cam, sh_degree=3)
    loss_r = (ref["rgb"] * w_rgb).mean() + (ref["depth"] * w_d).mean() + (ref["normal"] * w_n).mean()
    loss_r.backward()
    for k in ("rgb", "depth", "accumulation"):
        assert (out[k].detach().cpu() - ref[k].detach()).abs().max().item() < 1e-3, k
    # normal image: pixels with ~zero splatted normal are ill-conditioned under normalisation
    dn = (out["normal"].detach().cpu() - ref["normal"].detach()).abs()
    assert dn.mean().item() < 1e-4 and (dn > 1e-2).float().mean().item() < 1e-3
    assert np.array_equal(out["info"]["tiles_per_gauss"].cpu().numpy(), ref["info"]["tiles_per_gauss"].numpy())
    assert np.array_equal(out["info"]["flatten_ids"].cpu().numpy(), ref["info"]["flatten_ids"].numpy())
    assert np.array_equal(out["info"]["isect_ids"].cpu().numpy(), ref["info"]["isect_ids"].numpy())
    assert np.array_equal(out["info"]["isect_offsets"].cpu().numpy(), ref["info"]["isect_offsets"].numpy())
    # parameter gradients: per tensor, relative to that tensor's own largest gradient (E2E_RTOL, fp32 atomics)
    for k in gp:
        e = rel_err(gp[k].grad, cp[k].grad)
        assert e < E2E_RTOL, f"grad {k}: rel err {e}"
    # absgrad side output exists with the right shape
    assert out["xys"].absgrad.shape == (1, 1000, 2)


@pytest.mark.parametrize("H,W", [(64, 80), (37, 53), (800, 800)])
def test_fused_ssim_l1_loss(dev, H, W):
    """Row N2: fused (1-l)*L1 + l*(1-SSIM11) against the op-by-op torch statement of the same
    formula (itself pinned to torchmetrics' definition), forward and gradient."""
    from fusionsense_amd import losses
    g = torch.Generator().manual_seed(H)
    gt = torch.rand(H, W, 3, generator=g)
    pred = (gt + 0.2 * torch.randn(H, W, 3, generator=g)).clamp(0, 1)
    p_ref = pred.clone().double().requires_grad_(True)
    l_ref = losses.rgb_loss(p_ref, gt.double(), fused=False)
    l_ref.backward()
    p_g = pred.to(dev).requires_grad_(True)
    l_g = losses.rgb_loss(p_g, gt.to(dev), fused=True)
    (l_g * 3.0).backward()
    assert abs(l_g.item() - l_ref.item()) < 2e-5
    assert rel_err(p_g.grad / 3.0, p_ref.grad) < 1e-3


@pytest.mark.parametrize("mask_kind", ["ellipse", "all_zero", "one_pixel", "all_one"])
def test_masked_ssim_zero_tiles_take_the_short_way_to_the_same_bits(dev, mask_kind):
    """fsgs_ssim_l1_fwd_masked / _bwd_masked skip the image loads and the blur passes of tiles whose mask is all zero
    (forward: over the tile's halo; backward: over its own pixels).  The results must be those of the UNMASKED kernels on
    images multiplied by the mask beforehand — partial sums, the three derivative maps and mask * v_pred, bit for bit —
    whatever falls on a tile border (a single live pixel in a corner reaches four tiles through their halos)."""
    from fusionsense_amd._lib import load, ptr, stream_ptr
    lib = load()
    H, W = 150, 210  # 5 x 7 tiles of 32, ragged last row / column
    g = torch.Generator().manual_seed(31)
    pred, gt = torch.rand(H, W, 3, generator=g).to(dev), torch.rand(H, W, 3, generator=g).to(dev)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    if mask_kind == "ellipse":
        mask = (((xx - 80) / 60.0) ** 2 + ((yy - 70) / 45.0) ** 2 < 1.0).float()
    elif mask_kind == "all_zero":
        mask = torch.zeros(H, W)
    elif mask_kind == "all_one":
        mask = torch.ones(H, W)
    else:
        mask = torch.zeros(H, W)
        mask[64, 96] = 1.0  # the first pixel of tile (2, 3): inside the halos of (1, 2), (1, 3), (2, 2) too
    mask = mask.to(dev).contiguous()
    sp = stream_ptr(dev)
    n_blk = 3 * ((H + 31) // 32) * ((W + 31) // 32)
    v_loss = torch.tensor([0.7], device=dev)

    def run(p_, g_, m_):
        maps = [torch.full((3, H, W), float("nan"), device=dev) for _ in range(3)]
        sums = torch.full((n_blk, 2), float("nan"), device=dev)
        assert lib.fsgs_ssim_l1_fwd_masked(H, W, ptr(p_), ptr(g_), ptr(m_), ptr(maps[0]), ptr(maps[1]), ptr(maps[2]),
                                           ptr(sums), sp) == 0
        v = torch.full((H, W, 3), float("nan"), device=dev)
        assert lib.fsgs_ssim_l1_bwd_masked(H, W, ptr(p_), ptr(g_), ptr(m_), ptr(maps[0]), ptr(maps[1]), ptr(maps[2]),
                                           ptr(v_loss), 0.8 / (H * W * 3), -0.2 / ((H - 10) * (W - 10) * 3), ptr(v), 0,
                                           None, None, None, None, 0.0, None, sp) == 0
        torch.cuda.synchronize()
        return maps, sums, v
    maps_m, sums_m, v_m = run(pred, gt, mask)
    maps_u, sums_u, v_u = run((pred * mask[..., None]).contiguous(), (gt * mask[..., None]).contiguous(), None)
    assert torch.equal(sums_m, sums_u)
    for a, b in zip(maps_m, maps_u):
        assert torch.equal(a, b)
    assert torch.isfinite(v_m).all()
    assert torch.equal(v_m, v_u * mask[..., None])
    if mask_kind == "all_zero":
        assert float(v_m.abs().max()) == 0.0 and float(sums_m[:, 0].abs().max()) == 0.0
    # ... and the short way against the LONG way of the same instantiation: all-zero images under an all-one mask (every
    # tile live, every moment zero) must leave in every masked-out tile what the masked run left there
    zeros = torch.zeros(H, W, 3, device=dev)
    maps_z, sums_z, _ = run(zeros, zeros, torch.ones(H, W, device=dev))
    gy, gx = (H + 31) // 32, (W + 31) // 32
    n_short = 0
    for ty in range(gy):
        for tx in range(gx):
            if bool(mask[max(0, ty * 32 - 5):ty * 32 + 37, max(0, tx * 32 - 5):tx * 32 + 37].any()):
                continue
            n_short += 1
            for ch in range(3):
                blk = (ch * gy + ty) * gx + tx
                assert torch.equal(sums_m[blk], sums_z[blk])
            for a, b in zip(maps_m, maps_z):
                assert torch.equal(a[:, ty * 32:ty * 32 + 32, tx * 32:tx * 32 + 32], b[:, ty * 32:ty * 32 + 32, tx * 32:tx * 32 + 32])
    assert n_short == {"ellipse": 17, "all_zero": gy * gx, "one_pixel": gy * gx - 4, "all_one": 0}[mask_kind]


def test_fusion_aux_riders_equal_the_stand_alone_launches(dev):
    """fsgs_fusion_aux_loss_riders: the min-scale and touch-normal partial sums computed by extra workgroups of the aux
    launch are those of fsgs_min_scale_loss / fsgs_touch_normal_sqerr, and the aux partials are untouched by the riders
    (ragged sizes: rider blocks spill over several extra grid rows)."""
    import ctypes as C
    from fusionsense_amd._lib import load, ptr, stream_ptr
    lib = load()
    H, W, N, nt = 45, 70, 2001, 700   # gx = 3: 8 + 3 rider blocks over 4 extra rows
    g = torch.Generator().manual_seed(9)
    r = lambda *sh: torch.rand(*sh, generator=g).to(dev)  # noqa: E731
    depth, normal, image, sensor, ngt = r(H, W, 1) * 3, r(H, W, 3), r(H, W, 3), r(H, W) * 3, r(H, W, 3)
    mask = (r(H, W) > 0.2).float()
    log_scales = (torch.randn(N, 3, generator=g) - 4).to(dev)
    nw, tn = r(N, 3), r(nt, 3)
    idx = torch.randint(0, N, (nt,), generator=g).to(dev)
    w = (C.c_float * 7)(*[0.3, 0.2, 0.1, 0.05, 0.4, 0.1, 0.2])
    sp = stream_ptr(dev)
    P = lib.fsgs_fusion_aux_num_partials(H, W)
    a0, a1 = torch.zeros(P, 8, device=dev), torch.zeros(P, 8, device=dev)
    pm0, pm1 = torch.zeros((N + 255) // 256, 2, device=dev), torch.full(((N + 255) // 256, 2), -1.0, device=dev)
    pt0, pt1 = torch.zeros((nt + 255) // 256, 2, device=dev), torch.full(((nt + 255) // 256, 2), -1.0, device=dev)
    assert lib.fsgs_fusion_aux_loss(H, W, ptr(depth), ptr(normal), ptr(image), ptr(sensor), ptr(ngt), ptr(mask), 0.01, w,
                                    None, ptr(a0), None, None, 0, sp) == 0
    assert lib.fsgs_min_scale_loss(N, ptr(log_scales), 0.0, None, ptr(pm0), None, sp) == 0
    assert lib.fsgs_touch_normal_sqerr(nt, ptr(idx), ptr(nw), ptr(tn), ptr(pt0), sp) == 0
    assert lib.fsgs_fusion_aux_loss_riders(H, W, ptr(depth), ptr(normal), ptr(image), ptr(sensor), ptr(ngt), ptr(mask), 0.01,
                                           w, None, ptr(a1), None, None, 0, N, ptr(log_scales), ptr(pm1), nt, ptr(idx),
                                           ptr(nw), ptr(tn), ptr(pt1), sp) == 0
    assert torch.equal(a0, a1) and torch.equal(pm0, pm1) and torch.equal(pt0, pt1)
    assert float(pm0[:, 0].sum()) > 0 and float(pt0[:, 0].sum()) > 0
    # one rider only, and a missing buffer is refused
    pm1.fill_(-1.0)
    assert lib.fsgs_fusion_aux_loss_riders(H, W, ptr(depth), ptr(normal), ptr(image), ptr(sensor), ptr(ngt), ptr(mask), 0.01,
                                           w, None, ptr(a1), None, None, 0, N, ptr(log_scales), ptr(pm1), 0, None, None,
                                           None, None, sp) == 0
    assert torch.equal(pm0, pm1)
    assert lib.fsgs_fusion_aux_loss_riders(H, W, ptr(depth), ptr(normal), ptr(image), ptr(sensor), ptr(ngt), ptr(mask), 0.01,
                                           w, None, ptr(a1), None, None, 0, N, ptr(log_scales), None, 0, None, None, None,
                                           None, sp) != 0


@pytest.mark.parametrize("with_normal", [True, False])
def test_train_loss_single_node(dev, with_normal):
    """The whole config-#2 loss as one node (ssim+l1 on rgb, l1 on depth and normal) against torch."""
    from fusionsense_amd import losses
    from fusionsense_amd.ops import train_loss
    H, W = 45, 70
    g = torch.Generator().manual_seed(5)
    gt = {"rgb": torch.rand(H, W, 3, generator=g), "depth": torch.rand(H, W, 1, generator=g) * 3,
          "normal": torch.rand(H, W, 3, generator=g)}
    pred = {k: (v + 0.1 * torch.randn(v.shape, generator=g)) for k, v in gt.items()}
    pred["rgb"] = pred["rgb"].clamp(0, 1)
    ref = {k: v.clone().double().requires_grad_(True) for k, v in pred.items()}
    l_ref = losses.rgb_loss(ref["rgb"], gt["rgb"].double(), fused=False) \
        + 0.2 * (ref["depth"] - gt["depth"].double()).abs().mean()
    if with_normal:
        l_ref = l_ref + 0.1 * (ref["normal"] - gt["normal"].double()).abs().mean()
    l_ref.backward()
    gp = {k: v.to(dev).requires_grad_(True) for k, v in pred.items()}
    l_g = train_loss(gp["rgb"], gt["rgb"].to(dev), gp["depth"], gt["depth"].to(dev),
                     gp["normal"] if with_normal else None, gt["normal"].to(dev) if with_normal else None)
    (l_g * 2.0).backward()
    assert abs(l_g.item() - l_ref.item()) < 2e-5
    for k in ("rgb", "depth") + (("normal",) if with_normal else ()):
        assert rel_err(gp[k].grad / 2.0, ref[k].grad) < 1e-3, k


def test_legacy_pass_reuses_sorted_lists(dev):
    """The normal pass handed this frame's own xys/depths/radii skips its binning + sort; the
    image and gradients are identical to the full legacy binning, and any other tensor misses."""
    from fusionsense_amd import frame_cache
    from fusionsense_amd.fusion import render_fusionsense
    params, cam = scenes.cube_scene(1500, seed=9)
    outs = []
    for enabled in (True, False):
        frame_cache.enabled = enabled
        frame_cache.clear()
        h0, m0 = frame_cache.hits, frame_cache.misses
        gp = {k: v.to(dev).requires_grad_(True) for k, v in params.items()}
        out = render_fusionsense(gp, cam, sh_degree=3, device=dev)
        out["normal"].sum().backward()
        outs.append((out["normal"].detach().clone(), gp["quats"].grad.clone(), gp["opacities"].grad.clone()))
        if enabled:
            assert frame_cache.hits == h0 + 1, "reuse expected on the reference's own call pattern"
        else:
            assert frame_cache.hits == h0
    frame_cache.enabled = True
    assert out["info"]["legacy_rule_diff"] == 0, "cube scene: both bbox rules bin identically"
    assert torch.equal(outs[0][0], outs[1][0]), "forward is deterministic and list-identical"
    assert rel_err(outs[0][1], outs[1][1]) < 1e-4 and rel_err(outs[0][2], outs[1][2]) < 1e-4
    # a copy of the tensors (different storage) must not hit
    gp = {k: v.to(dev) for k, v in params.items()}
    out = render_fusionsense(gp, cam, sh_degree=3, device=dev)
    info = out["info"]
    from fusionsense_amd.legacy import rasterize_gaussians
    h0 = frame_cache.hits
    n_img = rasterize_gaussians(info["means2d"][0].clone(), info["depths"][0], info["radii"][0], info["conics"][0],
                                info["tiles_per_gauss"][0], torch.rand(1500, 3, device=dev),
                                torch.rand(1500, 1, device=dev), cam.height, cam.width, 16)
    assert frame_cache.hits == h0 and n_img.shape == (128, 128, 3)


@pytest.mark.parametrize("res", [(128, 128), (100, 75)])
def test_live_lists_equal_generic_walk(dev, res):
    """The culling/compaction fast path composites the same entries in the same order per pixel
    as the generic walk (outputs equal to fp32 rounding) and agrees on gradients up to atomic ordering;
    also on images whose size is not a multiple of the tile."""
    from fusionsense_amd import ops
    W, H = res
    params, cam = scenes.cube_scene(4000, seed=21)
    act = activated(params)
    # a few huge, a few nearly transparent and a few needle-like Gaussians
    act["scales"][:20] *= 15
    act["opacities"][20:60] = 0.004
    act["scales"][60:200, 0] *= 0.02
    viewmat, K = camera_mats(cam)
    K = K.clone()
    K[0, 0, 2], K[0, 1, 2] = W / 2, H / 2
    fr = oracle_frame(act, viewmat, K, W, H)
    g = torch.Generator().manual_seed(3)
    cols = torch.rand(1, 4000, 4, generator=g)
    opac = act["opacities"][None].contiguous()
    v_r, v_a = torch.randn(1, H, W, 4, generator=g), torch.randn(1, H, W, 1, generator=g)
    res_ = []
    for ids in (None, torch.from_numpy(fr["ids_s"]).to(dev)):
        ins = [t.to(dev).requires_grad_(True) for t in (fr["means2d"], fr["conics"], cols, opac)]
        out, alpha, last = ops._Rasterize.apply(*ins, None, torch.from_numpy(fr["offsets"]).to(dev),
                                                torch.from_numpy(fr["flat_s"]).to(dev), W, H, 16, True, ids)
        torch.autograd.backward([out, alpha], [v_r.to(dev), v_a.to(dev)])
        res_.append((out.detach(), alpha.detach(), last, [t.grad for t in ins], ins[0].absgrad))
    a, b = res_
    # same entries in the same order per pixel; the two kernels may contract FMAs differently
    assert (a[0] - b[0]).abs().max().item() < 2e-6 and (a[1] - b[1]).abs().max().item() < 2e-6
    assert (a[2] != b[2]).float().mean().item() < 1e-3
    # (the live backward restarts T / the colour suffix from the forward's saved segment states,
    # the generic one reconstructs them by division: same maths, different fp32 rounding)
    for ga, gb in zip(a[3], b[3]):
        assert rel_err(gb, ga) < 5e-4
    assert rel_err(b[4], a[4]) < 5e-4


def test_campos_kernel_matches_inverse(dev):
    from fusionsense_amd import ops
    g = torch.Generator().manual_seed(0)
    vm = torch.eye(4).repeat(7, 1, 1)
    vm[:, :3, :3] = torch.linalg.qr(torch.randn(7, 3, 3, generator=g))[0] * (0.5 + torch.rand(7, 1, 1, generator=g))
    vm[:, :3, 3] = torch.randn(7, 3, generator=g)
    ref = torch.linalg.inv(vm.double())[:, :3, 3]
    got = ops.campos_from_viewmats(vm.to(dev)).cpu().double()
    assert (got - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("mode", ["RGB+ED", "ED", "RGB+D", "RGB", "D"])
def test_render_modes_fused_vs_generic(dev, mode):
    """rasterization() in every render_mode: the live-list path (in-kernel expected-depth
    normalisation) against the generic kernels + gsplat's torch post-processing, and the oracle."""
    from fusionsense_amd import ops
    from fusionsense_amd.rendering import rasterization
    params, cam = scenes.cube_scene(800, seed=13)
    act = activated(params)
    viewmat, K = camera_mats(cam)
    g = torch.Generator().manual_seed(1)
    outs = []
    for live in (True, False):
        ops.USE_LIVE_LISTS = live
        leaves = {k: v.to(dev).requires_grad_(True) for k, v in act.items()}
        r, a, meta = rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                   leaves["colors"], viewmat.to(dev), K.to(dev), cam.width, cam.height,
                                   sh_degree=3, packed=False, render_mode=mode, absgrad=True)
        w = torch.rand(r.shape, generator=torch.Generator().manual_seed(2)).to(dev)
        ((r * w).sum() + a.sum()).backward()
        outs.append((r.detach().cpu(), a.detach().cpu(),
                     {k: v.grad.cpu() for k, v in leaves.items() if v.grad is not None}))
    ops.USE_LIVE_LISTS = True
    ref_leaves = {k: v.clone().requires_grad_(True) for k, v in act.items()}
    r_ref, a_ref, _ = R.rasterization(ref_leaves["means"], ref_leaves["quats"], ref_leaves["scales"],
                                      ref_leaves["opacities"], ref_leaves["colors"], viewmat, K, cam.width,
                                      cam.height, sh_degree=3, packed=False, render_mode=mode)
    for r, a, _ in outs:
        assert r.shape == r_ref.shape
        assert (r - r_ref.detach()).abs().max().item() < 1e-3
        assert (a - a_ref.detach()).abs().max().item() < 1e-4
    assert set(outs[0][2]) == set(outs[1][2])  # depth-only modes leave the SH coefficients untouched
    for k in outs[0][2]:
        assert rel_err(outs[0][2][k], outs[1][2][k]) < 2e-3, k


def test_fused_get_outputs_matches_unfused_and_oracle(dev, fwd_walk):
    """One-node get_outputs (single 7-channel walk, fused glue) against the op-by-op caller that
    goes through the drop-in surface, and against the CPU oracle, forward and gradients."""
    from fusionsense_amd.fused import render_fusionsense_fused
    from fusionsense_amd.fusion import render_fusionsense
    from oracle.fusion_ref import render_fusionsense as render_ref
    params, cam = scenes.cube_scene(1200, seed=31)
    g = torch.Generator().manual_seed(3)
    w_rgb, w_d, w_n, w_a = (torch.rand(128, 128, 3, generator=g), torch.rand(128, 128, 1, generator=g),
                            torch.rand(128, 128, 3, generator=g), torch.rand(128, 128, 1, generator=g))

    def run(fn, p, to=lambda t: t):
        out = fn(p, cam)
        loss = (out["rgb"] * to(w_rgb)).mean() + (out["depth"] * to(w_d)).mean() + \
            (out["normal"] * to(w_n)).mean() + (out["accumulation"] * to(w_a)).mean()
        loss.backward()
        return out

    pf = {k: v.to(dev).requires_grad_(True) for k, v in params.items()}
    of = run(lambda p, c: render_fusionsense_fused(p, c, sh_degree=3, device=dev), pf, lambda t: t.to(dev))
    pu = {k: v.to(dev).requires_grad_(True) for k, v in params.items()}
    ou = run(lambda p, c: render_fusionsense(p, c, sh_degree=3, device=dev), pu, lambda t: t.to(dev))
    pr = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    orf = run(lambda p, c: render_ref(p, c, sh_degree=3), pr)
    # expected depth = acc / alpha: at alpha ~ 1/255 one ulp of T moves it by ~5e-5
    for k, tol in (("rgb", 2e-5), ("depth", 3e-4), ("accumulation", 2e-5)):
        assert (of[k].detach() - ou[k].detach()).abs().max().item() < tol, k
        assert (of[k].detach().cpu() - orf[k].detach()).abs().max().item() < 1e-3, k
    dn = (of["normal"].detach() - ou["normal"].detach()).abs()
    assert dn.mean().item() < 1e-5 and (dn > 1e-2).float().mean().item() < 1e-3
    for k in pf:
        assert rel_err(pf[k].grad, pu[k].grad) < 2e-3, f"fused vs unfused grad {k}: {rel_err(pf[k].grad, pu[k].grad)}"
        assert rel_err(pf[k].grad, pr[k].grad) < 3e-3, f"fused vs oracle grad {k}: {rel_err(pf[k].grad, pr[k].grad)}"
    assert rel_err(of["info"].absgrad, ou["xys"].absgrad) < 2e-3
    assert torch.equal(of["radii"], ou["radii"])
    # the fused node bins only LIVE (Gaussian, tile) pairs; gsplat's rectangle count is still reported
    assert np.array_equal(of["info"]["tiles_per_gauss"].cpu().numpy(), orf["info"]["tiles_per_gauss"].numpy())
    n_live, n_full = of["info"]["flatten_ids"].numel(), orf["info"]["flatten_ids"].numel()
    assert 0 < n_live <= n_full
    live_ids = set(of["info"]["flatten_ids"].cpu().numpy().tolist())
    assert live_ids <= set(orf["info"]["flatten_ids"].numpy().tolist())


def test_fused_trainer_step_equals_unfused(dev, fwd_walk):
    """Two trainers from the same state, fused vs op-by-op caller: same loss, same gradients in the
    slab views, same parameters after the Adam step wherever the gradient is not numerically
    zero (Adam's g/sqrt(v) turns rounding noise on ~0 gradients into +-lr steps)."""
    from fusionsense_amd.trainer import SplatTrainer
    params, cam = scenes.cube_scene(900, seed=5)
    target = {"rgb": torch.rand(128, 128, 3).to(dev), "depth": torch.rand(128, 128, 1).to(dev),
              "normal": torch.rand(128, 128, 3).to(dev)}
    res = []
    for fused, aib in ((True, "0"), (False, "0"), (True, "1")):
        tr = SplatTrainer(params, dev, fused=fused)
        # "1": the Adam step is applied inside the per-Gaussian backward launch and the gradient slab is NOT filled
        # (trainer.adam_in_backward_mode, the default where the step qualifies); "0" keeps gradients + Adam launch
        tr.adam_in_backward_mode = aib
        before = {k: v.detach().clone() for k, v in tr.params.items()}
        loss = tr.train_step(cam, target)[0].item()
        tr.flush()  # (the fused trainer's feature update rides in the NEXT frame's count pass: land it)
        assert tr.adam_in_backward_steps == (1 if aib == "1" else 0)
        res.append((loss, {k: tr.slab.views[k].clone() for k in tr.params},
                    {k: (tr.params[k].detach() - before[k]) for k in tr.params}))
    assert abs(res[0][0] - res[1][0]) < 1e-5 * max(1.0, abs(res[1][0])) and res[2][0] == res[0][0]
    for k in res[0][1]:  # the step with Adam in the backward moves the parameters like the unfused step does
        gu = res[1][1][k]
        solid = gu.abs() > 1e-3 * gu.abs().max()
        assert (res[2][2][k][solid] - res[1][2][k][solid]).abs().max().item() < 1e-3 * tr.optim_cfg.lr[k] + 1e-9, k
    for k in res[0][1]:
        gf, gu = res[0][1][k], res[1][1][k]
        assert rel_err(gf, gu) < 5e-3, f"slab grad {k}: {rel_err(gf, gu)}"
        solid = gu.abs() > 1e-3 * gu.abs().max()
        assert solid.any()
        assert (res[0][2][k][solid] - res[1][2][k][solid]).abs().max().item() < 1e-3 * tr.optim_cfg.lr[k] + 1e-9, k


@pytest.mark.parametrize("case", ["random", "ties", "two_cameras", "bucket_1500", "bucket_2048", "bucket_3000", "bucket_4096",
                                  "bucket_5000", "bucket_20000", "single", "empty"])
def test_tile_sort_bit_exact(dev, case):
    """Partition-by-tile + per-tile LDS sort (csrc/tilesort.hip) against the radix sort of the same pairs
    and isect_offset_encode: sorted keys, payloads (ties in key -> ascending flatten id) and offsets must be
    identical, for buckets in each of the three size tiers (LDS small / LDS large / in place)."""
    from fusionsense_amd import ops
    g = torch.Generator().manual_seed(11)
    Cn, tw, th = (2, 7, 5) if case == "two_cameras" else (1, 9, 6)
    n_tiles = tw * th
    tb = ops.tile_bits(n_tiles)
    n = {"random": 30000, "ties": 20000, "two_cameras": 25000, "bucket_5000": 12000, "bucket_20000": 30000,
         "bucket_1500": 9000, "bucket_2048": 9000, "bucket_3000": 9000, "bucket_4096": 9000, "single": 1, "empty": 0}[case]
    tile = torch.randint(0, n_tiles, (n,), generator=g)
    cam = torch.randint(0, Cn, (n,), generator=g)
    depth = torch.rand(n, generator=g) * 10 + 0.1
    if case == "ties":
        depth = torch.round(depth)            # many equal (tile, depth) keys
    if case in ("bucket_1500", "bucket_2048", "bucket_3000", "bucket_4096"):  # the large LDS tier at and around its power-of-two sizes
        big = int(case.split("_")[1])
        tile[:] = torch.randint(0, n_tiles - 1, (n,), generator=g)
        tile[tile == 3] = 4
        tile[:big] = 3
        depth[:big // 2] = torch.round(depth[:big // 2] * 20) / 20   # ties inside the big bucket
    if case == "bucket_5000":
        tile[:5000] = 3
    if case == "bucket_20000":
        tile[:20000] = 7
        depth[:20000] = torch.round(depth[:20000] * 50) / 50
    gid = torch.randperm(max(n, 1), generator=g)[:n]            # flatten ids, all distinct
    order = torch.argsort(gid)                                  # emission order = ascending flatten id
    tile, cam, depth, gid = tile[order], cam[order], depth[order], gid[order]
    mask = torch.randint(1, 16, (n,), generator=g)
    key = (cam.long() << (32 + tb)) | (tile.long() << 32) | depth.view(torch.int32).long()
    pay = ((mask.long() << 28) | gid.long())
    pay = torch.where(pay >= 2**31, pay - 2**32, pay).to(torch.int32)
    ids_d, pay_d = key.to(dev), pay.to(dev)
    ids_s, pay_s, offs = ops.tile_sort(ids_d.clone(), pay_d.clone(), Cn, tw, th, want_ids=True)
    if n > 1:
        end_bit = 32 + tb + (ops.tile_bits(Cn) if Cn > 1 else 0)
        r_ids, r_pay = ops.sort_pairs(ids_d.clone(), pay_d.clone(), end_bit)
    else:
        r_ids, r_pay = ids_d, pay_d
    r_off = ops.isect_offset_encode(r_ids, Cn, tw, th)
    assert torch.equal(ids_s, r_ids)
    assert torch.equal(pay_s, r_pay)
    assert torch.equal(offs, r_off)


def _anisotropic_splats(seed, N, W, H):
    """Projected Gaussians as the projection would leave them: rotated, strongly elongated covariances (+ 0.3 blur),
    radius = ceil(3 sqrt(lambda_max)), opacities from below the 1/255 threshold up to 1 — the cases in which the
    enumeration rectangle of the direct binning is tighter than gsplat's square."""
    g = torch.Generator().manual_seed(seed)
    means2d = torch.rand(1, N, 2, generator=g) * torch.tensor([W, H]) * 1.3 - torch.tensor([W, H]) * 0.15
    s1 = torch.exp(torch.rand(1, N, generator=g) * 4.5 - 1.0)            # 0.37 .. 33 px
    s2 = s1 * torch.exp(-torch.rand(1, N, generator=g) * 3.5)            # up to 33 : 1
    th_ = torch.rand(1, N, generator=g) * math.pi
    c, s_ = torch.cos(th_), torch.sin(th_)
    a = c * c * s1**2 + s_ * s_ * s2**2 + 0.3
    b = c * s_ * (s1**2 - s2**2)
    d = s_ * s_ * s1**2 + c * c * s2**2 + 0.3
    det = a * d - b * b
    conics = torch.stack([d / det, -b / det, a / det], -1)
    mid = 0.5 * (a + d)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.01))
    radii = torch.ceil(3.0 * torch.sqrt(lam)).to(torch.int32)
    radii[torch.rand(1, N, generator=g) < 0.1] = 0
    opac = torch.exp(torch.rand(1, N, generator=g) * 6.5 - 6.5)           # 0.0015 .. 1
    depths = torch.rand(1, N, generator=g) * 5 + 0.2
    return means2d, radii, depths, conics, opac


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_direct_binning_of_anisotropic_and_faint_splats(dev, seed):
    """The direct binning enumerates the opacity-aware axis-aligned extent of a Gaussian (isect.hip) where the list
    chain walks gsplat's whole 3-sigma square: same live pairs, bit for bit, for rotated 30 : 1 ellipses, splats of
    60-pixel radius, opacities around 1 / 255 and centres outside the image."""
    from fusionsense_amd import ops
    W, H = 333, 190
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    a = [t.to(dev).contiguous() for t in _anisotropic_splats(seed, 20000, W, H)]
    tpg, ids, pay, offs = ops.bin_and_sort_live(a[0], a[1], a[2], a[3], a[4], tw, th)
    st = ops.bin_live_count_async(a[0], a[1], a[3], a[4], tw, th)
    d_tpg, _, d_pay, d_offs = ops.bin_live_finish(st, a[0], a[1], a[2], a[3], a[4], tw, th)
    assert torch.equal(d_tpg, tpg) and torch.equal(d_offs, offs) and torch.equal(d_pay, pay)
    assert pay.numel() > 20000 and int(tpg.sum()) > 1.3 * pay.numel(), "the case must contain dead pairs"


@pytest.mark.parametrize("case", ["one_camera", "two_cameras", "ties", "one_hot_tile", "one_depth_hot_tile", "tiny_image",
                                  "nothing_visible", "empty"])
def test_direct_binning_equals_list_chain(dev, case):
    """fsgs_bin_live_count + fsgs_bin_live_emit (pairs counted per tile and dropped straight into their
    tile's bucket) against fsgs_isect_count_live + emit_live + tile_sort on the same projected Gaussians:
    tiles_per_gauss, sorted payload (quadrant mask | flatten id) and isect_offsets bit for bit — several
    cameras, equal depths, one tile holding thousands of pairs (both sort tiers), an image smaller than a
    tile, nothing visible, N = 0."""
    from fusionsense_amd import ops
    g = torch.Generator().manual_seed(5)
    Cn = 2 if case == "two_cameras" else 1
    N = 0 if case == "empty" else 6000
    W, H = (12, 9) if case == "tiny_image" else (200, 136)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    means2d = torch.rand(Cn, N, 2, generator=g) * torch.tensor([W, H]) * 1.2 - torch.tensor([W, H]) * 0.1
    radii = torch.randint(0, 40, (Cn, N), generator=g, dtype=torch.int32)
    radii[torch.rand(Cn, N, generator=g) < 0.2] = 0
    if case == "nothing_visible":
        radii.zero_()
    depths = torch.rand(Cn, N, generator=g) * 5 + 0.2
    if case == "ties":
        depths = torch.round(depths * 2) / 2
    if case in ("one_hot_tile", "one_depth_hot_tile"):
        means2d[:, :5000] = torch.tensor([40.0, 40.0]) + torch.rand(Cn, 5000, 2, generator=g) * 4
        radii[:, :5000] = 6
    if case == "one_depth_hot_tile":  # a single depth value: the slab split cannot separate anything
        depths[:, :5000] = 1.25
    # conics of isotropic Gaussians whose 3-sigma radius is the bbox radius, some much tighter (culled pairs)
    sig = (radii.float().clamp(min=1) / 3) * (0.2 + 0.8 * torch.rand(Cn, N, generator=g))
    conics = torch.stack([1 / sig**2, 0.3 * torch.rand(Cn, N, generator=g) / sig**2, 1 / sig**2], -1)
    opac = torch.rand(Cn, N, generator=g)
    a = [t.to(dev).contiguous() for t in (means2d, radii, depths, conics, opac)]
    tpg, ids, pay, offs = ops.bin_and_sort_live(a[0], a[1], a[2], a[3], a[4], tw, th)
    for split in ("never", "always"):  # "always": buckets over 1024 pairs go through the depth-slab split
        ops.BIN_SPLIT = split
        try:
            st = ops.bin_live_count_async(a[0], a[1], a[3], a[4], tw, th)
            d_tpg, _, d_pay, d_offs = ops.bin_live_finish(st, a[0], a[1], a[2], a[3], a[4], tw, th)
        finally:
            ops.BIN_SPLIT = "auto"
        assert torch.equal(d_tpg, tpg)
        assert torch.equal(d_offs, offs)
        assert torch.equal(d_pay, pay), split
    if case not in ("nothing_visible", "empty"):
        assert pay.numel() > 0


def test_fused_adam_matches_torch(dev):
    """Row N1: one-launch Adam over several groups vs torch.optim.Adam (eps=1e-15 as in dn_config.py)."""
    from fusionsense_amd import ops
    g = torch.Generator().manual_seed(0)
    shapes = [(1000, 3), (1000, 4), (1000, 15, 3), (1000, 1), (7,)]
    lrs = [1.6e-4, 1e-3, 1.25e-4, 5e-2, 1e-2]
    ps = [torch.randn(*s, generator=g).to(dev) for s in shapes]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    opts = [torch.optim.Adam([r], lr=lr, eps=1e-15) for r, lr in zip(ref, lrs)]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    for step in range(1, 6):
        grads = [torch.randn(*s, generator=g).to(dev) * (10.0 ** -step) for s in shapes]
        for r, gr, o in zip(ref, grads, opts):
            r.grad = gr.clone()
            o.step()
        ops.adam_step_(ps, grads, ms, vs, lrs, step, 0.9, 0.999, 1e-15)
        for k, (p, r, o) in enumerate(zip(ps, ref, opts)):
            assert (p - r.detach()).abs().max().item() < 2e-6, step
            assert rel_err(ms[k], o.state[r]["exp_avg"]) < 1e-5
            assert rel_err(vs[k], o.state[r]["exp_avg_sq"]) < 1e-5


def test_folded_entry_points_equal_the_chains_they_replace(dev):
    """Every launch-saving entry point of the fused step against the calls it folds, on one frame of a cube scene
    (some Gaussians off screen or too small; covered and empty pixels), through the C-ABI:
      fsgs_project_bin_live_count  ==  fsgs_project_fwd_act + fsgs_bin_live_count           (projection to rounding,
                                                                                             counts bit for bit)
      fsgs_sh_fwd_pack             ==  fsgs_sh_fwd_split + fsgs_live_pack_normals           (to rounding)
      fsgs_epilogue_loss_fwd       ==  fsgs_epilogue_fwd + fsgs_aux_l1_fwd_bwd              (to rounding)
      fsgs_ssim_l1_bwd_combine     ==  fsgs_ssim_l1_bwd + fsgs_loss_combine                 (gradient bit for bit)
      fsgs_raster_bwd_quad_images  ==  fsgs_epilogue_bwd + fsgs_raster_bwd_quad             (float atomics: 1e-5)"""
    import ctypes as C
    from fusionsense_amd import ops
    from fusionsense_amd._lib import load, ptr, stream_ptr
    from fusionsense_amd.fused import _camera_on_device
    lib = load()
    sp = stream_ptr(dev)
    params, cam = scenes.cube_scene(3000, seed=7)
    P = {k: v.clone() for k, v in params.items()}
    P["means"][:40] *= 6.0        # some Gaussians off screen / behind the camera
    P["scales"][40:60] -= 3.0     # some below the radius threshold
    P["scales"][60:90] += 2.0     # some with a large footprint (gradient-line replicas, see the end)
    P = {k: v.to(dev).contiguous() for k, v in P.items()}
    N, K = P["means"].shape[0], 1 + P["features_rest"].shape[1]
    W, H = cam.width, cam.height
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    T = tw * th
    cd = _camera_on_device(cam, dev)
    f32 = dict(dtype=torch.float32, device=dev)

    def proj_bufs():
        return dict(scales_exp=torch.empty(N, 3, **f32), opac_sig=torch.empty(N, **f32),
                    radii=torch.empty(1, N, dtype=torch.int32, device=dev), means2d=torch.empty(1, N, 2, **f32),
                    depths=torch.empty(1, N, **f32), conics=torch.empty(1, N, 3, **f32))

    # ---- projection + count ------------------------------------------------------------------------------
    a, b = proj_bufs(), proj_bufs()
    opl_a, opl_b = P["opacities"].clone(), P["opacities"].clone()
    assert lib.fsgs_project_fwd_act(1, N, ptr(P["means"]), ptr(P["quats"]), ptr(P["scales"]), ptr(opl_a), 1, 0.1,
                                    ptr(cd["viewmat"]), ptr(cd["K"]), W, H, 0.3, 0.01, 1e10, 0.0, ptr(a["scales_exp"]),
                                    ptr(a["opac_sig"]), ptr(a["radii"]), ptr(a["means2d"]), ptr(a["depths"]),
                                    ptr(a["conics"]), sp) == 0
    st_a = ops.bin_live_count_async(a["means2d"], a["radii"], a["conics"], a["opac_sig"].view(1, N), tw, th)
    st_b = ops.project_bin_live_count_async(P["means"], P["quats"], P["scales"], opl_b, 0.1, cd["viewmat"], cd["K"],
                                            W, H, tw, th, b)
    torch.cuda.synchronize()
    assert torch.equal(opl_a, opl_b), "binary-opacity write on the parameter"
    for k in ("scales_exp", "opac_sig", "radii", "depths"):
        assert torch.equal(a[k], b[k]), k
    # the projection is the same statements inlined into two kernels: equal to rounding, not to the bit
    assert (a["means2d"] - b["means2d"]).abs().max().item() <= 1e-4
    assert rel_err(b["conics"], a["conics"]) < 1e-5
    assert torch.equal(st_a["tpg"], st_b["tpg"])
    assert int((a["radii"] > 0).sum()) > 1000
    # its counts are those of the stand-alone count pass on ITS projection, bit for bit
    st_c = ops.bin_live_count_async(b["means2d"], b["radii"], b["conics"], b["opac_sig"].view(1, N), tw, th)
    torch.cuda.synchronize()
    assert torch.equal(st_c["tpg"], st_b["tpg"]) and torch.equal(st_c["offsets"], st_b["offsets"])
    rc = ops.bin_live_finish(st_c, b["means2d"], b["radii"], b["depths"], b["conics"], b["opac_sig"].view(1, N), tw, th)
    rb = ops.bin_live_finish(st_b, b["means2d"], b["radii"], b["depths"], b["conics"], b["opac_sig"].view(1, N), tw, th)
    assert torch.equal(rc[2], rb[2])
    ra = ops.bin_live_finish(st_a, a["means2d"], a["radii"], a["depths"], a["conics"], a["opac_sig"].view(1, N), tw, th)
    payload, offsets = ra[2], st_a["offsets"]
    M = payload.numel()

    # ---- SH colours + packing ---------------------------------------------------------------------------
    n_cells = lib.fsgs_raster_quad_max_cells()
    colors = torch.empty(1, N, 4, **f32)
    pk_a, pk_b = torch.empty(N, 16, **f32), torch.empty(N, 16, **f32)
    nw_a, nw_b = torch.empty(N, 3, **f32), torch.empty(N, 3, **f32)
    mx_a, mx_b = torch.full((n_cells,), 7.0, **f32), torch.full((n_cells,), 7.0, **f32)
    assert lib.fsgs_sh_fwd_split(1, N, K, 3, ptr(P["means"]), ptr(cd["campos"]), ptr(P["features_dc"]),
                                 ptr(P["features_rest"]), ptr(a["radii"]), ptr(a["depths"]), ptr(colors), sp) == 0
    assert lib.fsgs_live_pack_normals(N, ptr(a["means2d"]), ptr(a["conics"]), ptr(colors), ptr(a["opac_sig"]),
                                      ptr(P["quats"]), ptr(P["scales"]), ptr(P["means"]), ptr(cd["c2w"]), ptr(pk_a),
                                      ptr(nw_a), ptr(mx_a), n_cells, sp) == 0
    assert lib.fsgs_sh_fwd_pack(N, K, 3, ptr(P["means"]), ptr(cd["campos"]), ptr(P["features_dc"]),
                                ptr(P["features_rest"]), ptr(a["radii"]), ptr(a["depths"]), ptr(a["means2d"]),
                                ptr(a["conics"]), ptr(a["opac_sig"]), ptr(P["quats"]), ptr(P["scales"]), ptr(cd["c2w"]),
                                ptr(pk_b), ptr(nw_b), ptr(mx_b), n_cells, sp) == 0
    assert torch.equal(pk_a[:, :8], pk_b[:, :8]) and torch.equal(mx_a, mx_b)  # copied fields: centre, opacity, conic
    # colours and normals are the same statements compiled into another kernel: equal to rounding
    assert (pk_a[:, 8:] - pk_b[:, 8:]).abs().max().item() <= 2e-6 and (nw_a - nw_b).abs().max().item() <= 2e-6

    # ---- the count pass with riders: SH forward + packing in its scan launch, an Adam step in its own ----------
    import ctypes as C
    c, opl_c = proj_bufs(), P["opacities"].clone()
    pk_c, nw_c, mx_c = torch.empty(N, 16, **f32), torch.empty(N, 3, **f32), torch.full((n_cells,), 7.0, **f32)
    gA = torch.Generator().manual_seed(11)
    prm = [torch.randn(5000, 3, generator=gA).to(dev), torch.randn(777, generator=gA).to(dev)]
    grd = [torch.randn(5000, 3, generator=gA).to(dev), torch.randn(777, generator=gA).to(dev)]
    mom = [torch.rand(5000, 3, generator=gA).to(dev) * 0.1, torch.rand(777, generator=gA).to(dev) * 0.1]
    var = [torch.rand(5000, 3, generator=gA).to(dev) * 0.01, torch.rand(777, generator=gA).to(dev) * 0.01]
    ref = [[t.clone() for t in grp] for grp in (prm, mom, var)]
    ops.adam_step_(ref[0], grd, ref[1], ref[2], [1e-2, 3e-3], 4)
    groups = ops.adam_groups(prm, grd, mom, var, [1e-2, 3e-3], 4)
    T_ = tw * th
    tpg_c = torch.empty(1, N, dtype=torch.int32, device=dev)
    offs_c = torch.empty(T_ + 1, dtype=torch.int32, device=dev)
    tbytes = lib.fsgs_bin_live_table_bytes(1, N, tw, th)
    table = torch.empty(tbytes, dtype=torch.uint8, device=dev)
    assert lib.fsgs_project_bin_live_count_sh_pack(
        N, ptr(P["means"]), ptr(P["quats"]), ptr(P["scales"]), ptr(opl_c), 1, 0.1, ptr(cd["viewmat"]), ptr(cd["K"]), W, H,
        0.3, 0.01, 1e10, 0.0, ptr(c["scales_exp"]), ptr(c["opac_sig"]), ptr(c["radii"]), ptr(c["means2d"]),
        ptr(c["depths"]), ptr(c["conics"]), tw, th, ptr(tpg_c), ptr(offs_c), ptr(table), tbytes, None, 3,
        ptr(cd["campos"]), ptr(P["features_dc"]), ptr(P["features_rest"]), ptr(cd["c2w"]), ptr(pk_c), ptr(nw_c),
        ptr(mx_c), n_cells, C.byref(groups), None, None, None, sp) == 0
    torch.cuda.synchronize()
    for k in ("scales_exp", "opac_sig", "radii", "depths", "means2d", "conics"):
        assert torch.equal(c[k], b[k]), k          # (the same kernel body as fsgs_project_bin_live_count)
    assert torch.equal(tpg_c, st_b["tpg"]) and torch.equal(offs_c, st_b["offsets"]) and torch.equal(opl_c, opl_b)
    # packed records from ITS projection: the copied fields exactly, colours / normals as the stand-alone launch
    pk_d, nw_d, mx_d = torch.empty(N, 16, **f32), torch.empty(N, 3, **f32), torch.full((n_cells,), 7.0, **f32)
    assert lib.fsgs_sh_fwd_pack(N, K, 3, ptr(P["means"]), ptr(cd["campos"]), ptr(P["features_dc"]),
                                ptr(P["features_rest"]), ptr(c["radii"]), ptr(c["depths"]), ptr(c["means2d"]),
                                ptr(c["conics"]), ptr(c["opac_sig"]), ptr(P["quats"]), ptr(P["scales"]), ptr(cd["c2w"]),
                                ptr(pk_d), ptr(nw_d), ptr(mx_d), n_cells, sp) == 0
    assert torch.equal(pk_c, pk_d) and torch.equal(nw_c, nw_d) and torch.equal(mx_c, mx_d)
    for got, want in zip((prm, mom, var), ref):     # the riding Adam step = fsgs_adam_step, bit for bit
        for x, y in zip(got, want):
            assert torch.equal(x, y)

    # ---- forward compositing (shared), then epilogue (+ aux loss) ---------------------------------------------
    cap = lib.fsgs_quad_stream_capacity(1, tw, th, M)
    records = torch.empty(4 * cap * 16, **f32)
    seg_state = torch.empty(4 * lib.fsgs_quad_seg_slots(1, tw, th, M) * 64 * 8, **f32)
    n_rec = torch.empty(4 * T, dtype=torch.int32, device=dev)
    render, alphas = torch.empty(1, H, W, 4, **f32), torch.empty(1, H, W, 1, **f32)
    last_ids = torch.empty(1, H, W, dtype=torch.int32, device=dev)
    extra = torch.empty(1, H, W, 3, **f32)
    assert lib.fsgs_raster_fwd_quad(1, 4, ptr(pk_a), ptr(payload), ptr(offsets), M, None, W, H, tw, th, 1, ptr(render),
                                    ptr(alphas), ptr(last_ids), ptr(records), ptr(n_rec), ptr(seg_state), ptr(extra),
                                    ptr(mx_a), None, 0, None, 0, 0, 0, 0, None, 0, None, None, 0, None, sp) == 0
    g = torch.Generator().manual_seed(3)
    bg = torch.tensor([1.0, 0.5, 0.25], device=dev)
    depth_gt, normal_gt = torch.rand(H, W, 1, generator=g).to(dev), torch.rand(H, W, 3, generator=g).to(dev)
    seed = torch.tensor(0.75, device=dev)
    Pn = H * W
    gd, gn = 0.2 / Pn, 0.1 / (3 * Pn)

    def imgs():
        return torch.empty(H, W, 3, **f32), torch.empty(H, W, 1, **f32), torch.empty(H, W, 3, **f32)
    rgb_a, dep_a, nrm_a = imgs()
    rgb_b, dep_b, nrm_b = imgs()
    part_a, part_b = torch.empty((Pn + 255) // 256, 2, **f32), torch.empty((Pn + 255) // 256, 2, **f32)
    vd_a, vd_b = torch.empty(H, W, 1, **f32), torch.empty(H, W, 1, **f32)
    vn_a, vn_b = torch.empty(H, W, 3, **f32), torch.empty(H, W, 3, **f32)
    assert lib.fsgs_epilogue_fwd(Pn, ptr(render), ptr(alphas), ptr(extra), ptr(bg), None, -n_cells, ptr(mx_a), ptr(rgb_a),
                                 ptr(dep_a), ptr(nrm_a), sp) == 0
    assert lib.fsgs_aux_l1_fwd_bwd(Pn, ptr(dep_a), ptr(depth_gt), ptr(nrm_a), ptr(normal_gt), ptr(part_a), ptr(seed), gd,
                                   gn, ptr(vd_a), ptr(vn_a), sp) == 0
    assert lib.fsgs_epilogue_loss_fwd(Pn, ptr(render), ptr(alphas), ptr(extra), ptr(bg), ptr(mx_a), n_cells, ptr(rgb_b),
                                      ptr(dep_b), ptr(nrm_b), ptr(depth_gt), ptr(normal_gt), ptr(seed), gd, gn,
                                      ptr(part_b), ptr(vd_b), ptr(vn_b), sp) == 0
    # (another instantiation of the same statements: images equal to rounding; the gradient images are signs, so
    # they can differ only where prediction and target agree to the last bit)
    for x, y in ((rgb_a, rgb_b), (dep_a, dep_b), (nrm_a, nrm_b)):
        assert (x - y).abs().max().item() <= 2e-6
    assert rel_err(part_b, part_a) < 1e-5
    for x, y in ((vd_a, vd_b), (vn_a, vn_b)):
        assert float((x != y).float().mean()) < 1e-3
    assert float(alphas.max()) > 0.5 and bool((alphas == 0).any()), "covered and empty pixels (the depth fill path)"

    # ---- SSIM backward (+ loss combine) ------------------------------------------------------------------
    rgb_gt = torch.rand(H, W, 3, generator=g).to(dev)
    maps = torch.empty(3, H, W, 3, **f32)
    sums = torch.empty(lib.fsgs_ssim_l1_num_partials(H, W), 2, **f32)
    assert lib.fsgs_ssim_l1_fwd(H, W, ptr(rgb_a), ptr(rgb_gt), maps[0].data_ptr(), maps[1].data_ptr(),
                                maps[2].data_ptr(), ptr(sums), sp) == 0
    g1, gs = 0.8 / (3 * Pn), -0.2 / (3.0 * (H - 10) * (W - 10))
    vr_a, vr_b = torch.empty(H, W, 3, **f32), torch.empty(H, W, 3, **f32)
    assert lib.fsgs_ssim_l1_bwd(H, W, ptr(rgb_a), ptr(rgb_gt), maps[0].data_ptr(), maps[1].data_ptr(),
                                maps[2].data_ptr(), ptr(seed), g1, gs, ptr(vr_a), sp) == 0
    loss_a = ops.loss_combine([sums, part_a], [(g1, gs), (gd, gn)], 0.2)
    loss_b = torch.empty((), **f32)
    VP = C.c_void_p * 2
    rows = (C.c_int64 * 2)(sums.shape[0], part_a.shape[0])
    w = (C.c_float * 4)(g1, gs, gd, gn)
    assert lib.fsgs_ssim_l1_bwd_combine(H, W, ptr(rgb_a), ptr(rgb_gt), maps[0].data_ptr(), maps[1].data_ptr(),
                                        maps[2].data_ptr(), ptr(seed), g1, gs, ptr(vr_b), 2,
                                        VP(sums.data_ptr(), part_a.data_ptr()), rows, w, 0.2, ptr(loss_b), sp) == 0
    assert torch.equal(vr_a, vr_b)
    assert abs(float(loss_a) - float(loss_b)) <= 1e-6 * abs(float(loss_a))  # (f64 sums in a different order)

    # ---- backward compositing from the image gradients -----------------------------------------------------------
    v_alpha = torch.randn(H, W, 1, generator=g).to(dev) * 1e-3
    v_render, v_alphas, v_extra = torch.empty(1, H, W, 4, **f32), torch.empty(1, H, W, 1, **f32), torch.empty(1, H, W, 3, **f32)
    assert lib.fsgs_epilogue_bwd(Pn, ptr(render), ptr(alphas), ptr(extra), ptr(bg), ptr(vr_a), ptr(vd_a), ptr(vn_a),
                                 ptr(v_alpha), ptr(v_render), ptr(v_alphas), ptr(v_extra), sp) == 0
    acc_a, acc_b = torch.zeros(N, 16, **f32), torch.zeros(N, 16, **f32)
    assert lib.fsgs_raster_bwd_quad(1, 4, ptr(records), ptr(n_rec), ptr(offsets), M, None, W, H, tw, th, 1, ptr(render),
                                    ptr(alphas), ptr(last_ids), ptr(v_render), ptr(v_alphas), ptr(seg_state), 1,
                                    ptr(extra), ptr(v_extra), ptr(acc_a), 0, None, None, 0, None, sp) == 0
    assert lib.fsgs_raster_bwd_quad_images(ptr(records), ptr(n_rec), ptr(offsets), M, W, H, tw, th, ptr(render),
                                           ptr(alphas), ptr(last_ids), ptr(extra), ptr(bg), ptr(vr_a), ptr(vd_a),
                                           ptr(vn_a), ptr(v_alpha), ptr(seg_state), 1, ptr(acc_b), 0, 7, None, None, 0,
                                           None, sp) == 0
    assert float(acc_a.abs().max()) > 0
    assert rel_err(acc_b, acc_a) < 1e-5
    # gradient-line replicas (common.h grad_spread): large Gaussians spread their atomics over 4 lines N rows apart,
    # small ones keep to line 0; the lines sum to the unspread result
    LN = lib.fsgs_grad_replica_lines()  # 4 lines by footprint tier 1, all LN (16) by tier 2
    assert LN >= 4
    acc_r = torch.zeros(LN * N, 16, **f32)
    assert lib.fsgs_raster_bwd_quad_images(ptr(records), ptr(n_rec), ptr(offsets), M, W, H, tw, th, ptr(render),
                                           ptr(alphas), ptr(last_ids), ptr(extra), ptr(bg), ptr(vr_a), ptr(vd_a),
                                           ptr(vn_a), ptr(v_alpha), ptr(seg_state), 1, ptr(acc_r), N, -1, None, None, 0,
                                           None, sp) == 0
    lines = acc_r.view(LN, N, 16)
    assert rel_err(lines.sum(0), acc_a) < 1e-5
    used = lines[1:].abs().sum((0, 2)) > 0
    assert bool(used.any()) and not bool(used.all())


def test_finite_differences_of_the_hip_path(dev):
    """A check of the HIP forward + backward that does NOT go through the recalled oracle: on an 8-Gaussian scene the
    analytic parameter gradients of a smooth image functional equal central finite differences of the HIP forward
    itself, along random directions in every parameter group (fp32: step 2e-3, agreement to 2 % of |g.d| + floor)."""
    from fusionsense_amd.fused import render_fusionsense_fused
    g = torch.Generator().manual_seed(11)
    n = 8
    # Gaussians much LARGER than the 32x32 image and moderately opaque: every (pixel, Gaussian) pair stays far above the
    # alpha >= 1/255 cut and below the T <= 1e-4 stop, so the rendered functional is smooth (those hard thresholds are
    # part of the algorithm, and finite differences across them do not estimate the gradient the backward defines)
    params = dict(means=0.25 * torch.randn(n, 3, generator=g), scales=math.log(1.2) + 0.15 * torch.randn(n, 3, generator=g),
                  quats=torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1),
                  features_dc=torch.randn(n, 3, generator=g), features_rest=0.2 * torch.randn(n, 15, 3, generator=g),
                  opacities=-1.1 + 0.3 * torch.randn(n, 1, generator=g))
    cam = scenes.Camera(scenes.look_at_c2w(torch.tensor([0.0, 0.0, 4.0]), torch.zeros(3), up=(0.0, 1.0, 0.0)),
                        70.0, 70.0, 16.0, 16.0, 32, 32)
    w = {k: torch.rand(32, 32, c, generator=g).to(dev) for k, c in (("rgb", 3), ("depth", 1), ("normal", 3))}

    with torch.no_grad():  # pixels well inside the splats at the base point: away from the detached max-depth fill of
        # empty pixels (dn_model.py:609-612) and from 0/0 normals, where the functional is not differentiable by design
        base = render_fusionsense_fused({k: v.to(dev) for k, v in params.items()}, cam, sh_degree=3, device=dev)
        inside = (base["accumulation"] > 0.3).float()
    assert float(inside.mean()) > 0.9

    def functional(p, keys, grad=False):
        pp = {k: v.to(dev).requires_grad_(grad) for k, v in p.items()}
        o = render_fusionsense_fused(pp, cam, sh_degree=3, device=dev)
        val = sum((o[k] * w[k] * inside).mean() for k in keys)
        return val, pp

    eps = 2e-3
    for k in params:
        # the normal plane is rasterized at DETACHED centres (xys.detach(), dn_model.py:638) and its per-Gaussian
        # normals use detached view directions: its dependence on the means is not differentiated, by design
        keys = ("rgb", "depth") if k == "means" else ("rgb", "depth", "normal")
        val, pp = functional(params, keys, True)
        val.backward()

        def functional_k(p, _keys=keys):
            return functional(p, _keys)

        for trial in range(3):
            d = torch.randn(params[k].shape, generator=g)
            d = d / d.norm()
            plus = {**params, k: params[k] + eps * d}
            minus = {**params, k: params[k] - eps * d}
            with torch.no_grad():
                fd = (float(functional_k(plus)[0].double()) - float(functional_k(minus)[0].double())) / (2 * eps)
            an = float((pp[k].grad.cpu().double() * d.double()).sum())
            assert abs(fd - an) <= 2e-2 * abs(an) + 3e-5, (k, trial, fd, an)

    # the NORMAL PLANE alone (the E = 3 channels of the compositing walk: its gradient reaches conics and opacities —
    # i.e. scales, quaternions, opacity logits — and its own colours, the per-Gaussian normals, never the centres):
    for k in ("scales", "quats", "opacities"):
        val, pp = functional(params, ("normal",), True)
        val.backward()
        for trial in range(3):
            d = torch.randn(params[k].shape, generator=g)
            d = d / d.norm()
            with torch.no_grad():
                fd = (float(functional({**params, k: params[k] + eps * d}, ("normal",))[0].double()) -
                      float(functional({**params, k: params[k] - eps * d}, ("normal",))[0].double())) / (2 * eps)
            an = float((pp[k].grad.cpu().double() * d.double()).sum())
            assert abs(fd - an) <= 2e-2 * abs(an) + 3e-5, ("normal plane", k, trial, fd, an)
    # ... and the SH coefficients band by band (degree 1: rows 0..2 of features_rest, degree 2: 3..7, degree 3: 8..14),
    # so that a wrong basis constant in one band cannot hide behind the others
    val, pp = functional(params, ("rgb",), True)
    val.backward()
    for lo, hi in ((0, 3), (3, 8), (8, 15)):
        d = torch.zeros(params["features_rest"].shape)
        d[:, lo:hi] = torch.randn(n, hi - lo, 3, generator=g)
        d = d / d.norm()
        with torch.no_grad():
            fd = (float(functional({**params, "features_rest": params["features_rest"] + eps * d}, ("rgb",))[0].double()) -
                  float(functional({**params, "features_rest": params["features_rest"] - eps * d}, ("rgb",))[0].double())) / (2 * eps)
        an = float((pp["features_rest"].grad.cpu().double() * d.double()).sum())
        assert abs(an) > 1e-6 and abs(fd - an) <= 2e-2 * abs(an) + 3e-5, ("SH band", lo, hi, fd, an)


def test_sh_colours_with_differentiable_camera_pose(dev):
    """rasterization(sh_degree=3) with viewmats.requires_grad (nerfstudio's camera optimiser; off in FusionSense):
    the view-matrix gradient — projection VJP plus the SH view-direction share through campos = inverse(viewmat)[:3,3]
    — against fp64 autograd of the oracle."""
    from fusionsense_amd.rendering import rasterization
    params, cam = scenes.cube_scene(600, seed=6)
    act = activated(params)
    viewmat, K = camera_mats(cam)
    g = torch.Generator().manual_seed(1)
    w = torch.rand(1, cam.height, cam.width, 4, generator=g)
    vm64 = viewmat.double().requires_grad_(True)
    r64, a64, _ = R.rasterization(act["means"].double(), act["quats"].double(), act["scales"].double(),
                                  act["opacities"].double(), act["colors"].double(), vm64, K.double(), cam.width, cam.height,
                                  sh_degree=3, render_mode="RGB+ED", packed=False)
    (r64 * w.double()).sum().backward()
    vm = viewmat.to(dev).requires_grad_(True)
    r, a, _ = rasterization(act["means"].to(dev), act["quats"].to(dev), act["scales"].to(dev), act["opacities"].to(dev),
                            act["colors"].to(dev), vm, K.to(dev), cam.width, cam.height, sh_degree=3,
                            render_mode="RGB+ED", packed=False)
    (r * w.to(dev)).sum().backward()
    assert rel_err(r, r64) < 2e-4
    assert vm.grad is not None and rel_err(vm.grad[:, :3], vm64.grad[:, :3]) < 5e-3, rel_err(vm.grad[:, :3], vm64.grad[:, :3])
    # and the SH share is really there: without it the gradient differs
    vm2 = viewmat.to(dev).requires_grad_(True)
    r2, _, _ = rasterization(act["means"].to(dev), act["quats"].to(dev), act["scales"].to(dev), act["opacities"].to(dev),
                             act["colors"].to(dev), vm2, K.to(dev), cam.width, cam.height, sh_degree=0,
                             render_mode="RGB+ED", packed=False)
    assert r2.shape == r.shape


def test_rasterization_without_host_wait_is_bit_identical_and_survives_overflow(dev):
    """rasterization() (the drop-in surface) sizes its live lists from earlier frames of the same shape and looks at
    the frame's live total only after the compositing forward is enqueued (rendering.NO_WAIT): same images, lists and
    gradients as the waiting path, the legacy normal pass reuses the capacity-sized lists, and a frame that exceeds a
    (deliberately tiny) capacity is binned and composited once more with exact sizes."""
    from fusionsense_amd import rendering
    from fusionsense_amd.fusion import render_fusionsense
    params, cam = scenes.cube_scene(3000, seed=12)

    def run(no_wait, poison=None):
        rendering.NO_WAIT = no_wait
        rendering._LIVE_CAPS.clear()
        outs = []
        for it in range(3):
            if poison is not None and it == 2:
                poison_capacity(rendering._LIVE_CAPS, poison)
            gp = {k: v.to(dev).requires_grad_(True) for k, v in params.items()}
            o = render_fusionsense(gp, cam, sh_degree=3, device=dev)
            (o["rgb"].sum() + o["depth"].sum() + o["normal"].sum()).backward()
            outs.append((o, {k: p.grad.clone() for k, p in gp.items()}))
        return outs

    try:
        ref = run(False)
        assert not rendering._LIVE_CAPS
        nw = run(True)
        assert rendering._LIVE_CAPS and rendering.live_overflows == 0
        n0 = rendering.live_overflows
        ov = run(True, poison=64)
        assert rendering.live_overflows == n0 + 1
    finally:
        rendering.NO_WAIT = True
        rendering._LIVE_CAPS.clear()
    for other in (nw, ov):
        for (o0, g0), (o1, g1) in zip(ref, other):
            for k in ("rgb", "depth", "normal", "accumulation"):
                assert torch.equal(o0[k], o1[k]), k
            assert torch.equal(o0["info"]["live_payload"], o1["info"]["live_payload"])
            assert torch.equal(o0["info"]["live_offsets"], o1["info"]["live_offsets"])
            assert torch.equal(o0["info"]["last_ids"], o1["info"]["last_ids"])
            assert torch.equal(o0["info"]["isect_ids"], o1["info"]["isect_ids"])  # gsplat's full lists, built lazily
            for k in g0:
                assert rel_err(g1[k], g0[k]) < 2e-4, (k, rel_err(g1[k], g0[k]))  # (float atomics in another order)
            assert rel_err(o1["xys"].absgrad, o0["xys"].absgrad) < 2e-4
