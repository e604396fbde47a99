"""BASELINE configs #4 / #5 on the scene bench.py QUOTES them on (``scenes.surface_dense_scene`` seen from
``scenes.ring_cameras``: Gaussians on surfaces, lists of several hundred walked entries per pixel, tiles of more than 8192
live pairs that go through the depth-slab sorts), on both walks of the forward compositing and through the route the bench
takes (``SplatTrainer.train_step``: tape-free step, no-wait list capacity, occlusion cuts from the view's previous frame):

* at FULL size (6 M / 10 M Gaussians, 1920x1080) through size-independent properties — lists sorted by (tile, depth, id)
  with offsets at the tile boundaries, equal bit for bit to the independent radix chain, ``last_ids`` inside the pixel's own
  tile list, bit-reproducible forward in range, the cut frame equal to the uncut frame, backward linear and reproducible,
  allocation-free steps, half storage = the fp32 path on the rounded values;
* against the CPU oracle on a window of the same view with ALL of the scene's Gaussians (the pattern of
  ``test_config2_against_oracle_at_its_own_size``): integer outputs, images, every parameter gradient, with every
  mismatch required to be an fp32 threshold case.

All through the C-ABI ("Next round" item 1)."""
import math

import pytest
import torch

from fusionsense_amd import scenes
from helpers import crop_camera, rel_err

pytestmark = pytest.mark.gpu

N4, N5, W4, H4 = 6_000_000, 10_000_000, 1920, 1080
TW, TH = math.ceil(W4 / 16), math.ceil(H4 / 16)
# the oracle's window: 128x128 pixels below the image centre, where the ring cameras see ground and boxes (every pixel
# saturates: mean accumulation 0.9998, ~280 k list entries over 64 tiles — the depth-slab sorts run here too)
CROP, CROP_SHIFT = 128, (0, 60)


def _ring():
    return scenes.ring_cameras(8, width=W4, height_px=H4, focal=1500.0)


@pytest.fixture(scope="module")
def surface4(dev):
    params = scenes.surface_dense_scene(N4, seed=0)
    return {k: v.to(dev) for k, v in params.items()}, _ring()


def _targets(dev, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    return {"rgb": torch.rand(H, W, 3, generator=g).to(dev), "depth": (torch.rand(H, W, 1, generator=g) * 8.0 + 4.0).to(dev),
            "normal": torch.rand(H, W, 3, generator=g).to(dev)}


def _stats_trainer(dev, params, n_views, **kw):
    from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig
    from fusionsense_amd.trainer import SplatTrainer
    st = DensifyStrategy(SplatfactoConfig(), num_train_data=n_views, stats_only=True)
    return SplatTrainer(params, dev, sh_degree=3, strategy=st, **kw), st


def _check_lists(dev, params, info, W, H, min_pairs, half=None):
    """Size-independent properties of a frame's live lists + their equality with the independent radix chain."""
    from fusionsense_amd import ops
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    pay, offs = info.payload, info.isect_offsets.flatten().long()
    M = pay.numel()
    assert M > min_pairs
    gid = (pay & 0x0FFFFFFF).long()
    assert bool(((pay >> 28) & 0xF != 0).all()) and bool((info.radii[0][gid] > 0).all())
    ends = torch.cat([offs[1:], torch.tensor([M], device=dev)])
    assert bool((ends >= offs).all()) and int(offs[0]) == 0
    tile_of = torch.searchsorted(offs.contiguous(), torch.arange(M, device=dev), right=True) - 1
    depth_bits = info.depths[0][gid].view(torch.int32).long()
    same_tile = tile_of[1:] == tile_of[:-1]
    assert bool((depth_bits[1:] >= depth_bits[:-1])[same_tile].all()), "depth order inside a tile"
    tie = same_tile & (depth_bits[1:] == depth_bits[:-1])
    assert bool((gid[1:] > gid[:-1])[tie].all()), "ties keep ascending Gaussian id"
    tx, ty = tile_of % tw, tile_of // tw
    mx, my, r = info.means2d[0][gid, 0], info.means2d[0][gid, 1], info.radii[0][gid].float()
    assert bool(((mx + r > tx * 16) & (mx - r < tx * 16 + 16) & (my + r > ty * 16) & (my - r < ty * 16 + 16)).all())
    # the independent radix route over the same projected Gaussians: the same lists, bit for bit
    if half is not None:
        opac = torch.sigmoid(half["opacities"].float()).view(1, -1).contiguous()
    else:
        opac = torch.sigmoid(params["opacities"]).view(1, -1).contiguous()
    st = ops.isect_count_live_async(info.means2d, info.radii, info.conics, opac, tw, th)
    ops.USE_TILE_SORT = False
    try:
        _, _, r_pay, r_offs = ops.isect_finish_live(st, info.means2d, info.radii, info.depths, info.conics, opac, tw, th)
    finally:
        ops.USE_TILE_SORT = True
    assert torch.equal(r_pay, pay) and torch.equal(r_offs, info.isect_offsets)
    n_tile = ends - offs
    return M, n_tile


def _check_last_ids(dev, out, W, H):
    tw = math.ceil(W / 16)
    info = out["info"]
    offs = info.isect_offsets.flatten().long()
    ends = torch.cat([offs[1:], torch.tensor([info.payload.numel()], device=dev)])
    last = info.last_ids[0].long()
    t = (torch.arange(H, device=dev)[:, None] // 16) * tw + torch.arange(W, device=dev)[None, :] // 16
    hit = out["accumulation"][..., 0] > 0
    assert bool(((last >= offs[t]) & (last < ends[t]))[hit].all())


def test_config4_surface_scene_lists_and_forward(dev, surface4, fwd_walk):
    """6 M Gaussians on surfaces at 1080p, both forward walks: the dense binning route (depth slabs; some tiles hold more
    than 8192 live pairs) leaves lists that are sorted, consistent and bit-identical to the radix chain; the forward is
    bit-reproducible and in range, last_ids index the pixel's own tile list, pixels saturate (a surface scene: more
    than half of the image — everything below the horizon — ends opaque)."""
    from fusionsense_amd import ops
    from fusionsense_amd.fused import render_fusionsense_fused
    params, cams = surface4
    with torch.no_grad():
        a = render_fusionsense_fused(params, cams[0], sh_degree=3, device=dev)
        b = render_fusionsense_fused(params, cams[0], sh_degree=3, device=dev)
    info = a["info"]
    assert info.isect_offsets.shape == (1, TH, TW)
    for k in ("rgb", "depth", "normal", "accumulation"):
        assert torch.equal(a[k], b[k]) and bool(torch.isfinite(a[k]).all()), k
    assert torch.equal(info.payload, b["info"].payload) and torch.equal(info.last_ids, b["info"].last_ids)
    assert 0.0 <= float(a["rgb"].min()) and float(a["rgb"].max()) <= 1.0 and float(a["accumulation"].max()) <= 1.0
    acc = a["accumulation"][..., 0]
    assert float((acc > 0.99).float().mean()) > 0.5, "a surface scene: most pixels below the horizon saturate"
    M, n_tile = _check_lists(dev, params, info, W4, H4, min_pairs=5_000_000)
    assert M > 1024 * TW * TH // 2 and not ops.use_tile_sort(M, TW * TH), "the scene must take the dense (slab) sorts"
    assert int(n_tile.max()) > 8192, "some tiles exceed even the large LDS tier before the split"
    _check_last_ids(dev, a, W4, H4)


def test_config4_surface_scene_bench_route(dev, surface4, fwd_walk):
    """What bench.py --config 4 runs: SplatTrainer.train_step on the ring views — tape-free step, list buffers from the
    capacity estimate, occlusion cuts from the view's previous frame.  The cut frame's images equal the uncut frame's
    bit for bit (parameters held: optimizer_step=False), gradients to the atomics' reordering; lists of the cut frame
    are a depth prefix of the uncut lists per tile; then real steps are finite, move the parameters, allocate nothing."""
    from fusionsense_amd.fused import render_fusionsense_fused
    params, cams = surface4
    views = [cams[0], cams[3]]
    tr, st = _stats_trainer(dev, params, len(views))
    tgt = [_targets(dev, H4, W4, 5 + i) for i in range(2)]
    with torch.no_grad():
        ref = render_fusionsense_fused(params, views[0], sh_degree=3, device=dev)
    l0, o0 = tr.train_step(views[0], tgt[0], optimizer_step=False)  # first visit: no cuts yet
    g0 = {k: v.clone() for k, v in tr.slab.views.items()}
    img0 = {k: o0[k].clone() for k in ("rgb", "depth", "normal", "accumulation")}
    pay0, offs0 = o0["info"].payload.clone(), o0["info"].isect_offsets.flatten().long().clone()
    for k in img0:  # the trainer's frame IS the fused node's frame (its epilogue is the loss-fused instantiation: the
        # normalisation of the normal image may round differently by an ulp)
        assert torch.equal(img0[k], ref[k]) if k != "normal" else float((img0[k] - ref[k]).abs().max()) <= 2e-7, k
    l1, o1 = tr.train_step(views[0], tgt[0], optimizer_step=False)  # second visit: binned with the view's cuts
    assert tr.cut_frames >= 1, "a 6 M-Gaussian trainer bins a revisited view with occlusion cuts"
    for k in img0:
        assert torch.equal(o1[k], img0[k]), k
    assert float(l1) == float(l0)
    for k in g0:
        assert rel_err(tr.slab.views[k], g0[k]) < 5e-4, k
    # per tile the cut list is a prefix of the uncut list
    pay1, offs1 = o1["info"].payload, o1["info"].isect_offsets.flatten().long()
    M0, M1 = pay0.numel(), pay1.numel()
    assert M1 <= M0
    n1 = torch.cat([offs1[1:], torch.tensor([M1], device=dev)]) - offs1
    n0 = torch.cat([offs0[1:], torch.tensor([M0], device=dev)]) - offs0
    assert bool((n1 <= n0).all())
    tile_of = torch.searchsorted(offs1.contiguous(), torch.arange(M1, device=dev), right=True) - 1
    src = offs0[tile_of] + (torch.arange(M1, device=dev) - offs1[tile_of])
    assert torch.equal(pay1, pay0[src])
    _check_last_ids(dev, o1, W4, H4)
    # real steps
    for c, t in zip(views, tgt):
        tr.train_step(c, t, optimizer_step=False)
    tr.train_step(views[0], tgt[0])
    tr.train_step(views[1], tgt[1])
    torch.cuda.synchronize()
    n_alloc = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    p0 = tr.params["means"].data.clone()
    for s in range(4):
        loss, _ = tr.train_step(views[s % 2], tgt[s % 2])
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - n_alloc <= 1, "a step must not allocate device memory"
    assert math.isfinite(float(loss)) and not torch.equal(p0, tr.params["means"].data)
    assert st.vis_counts is not None and float(st.vis_counts.max()) >= 2.0


def test_config4_surface_scene_backward_linear(dev, surface4, fwd_walk):
    """The VJP on the surface scene (lists of hundreds of walked entries per pixel, union walks, replica lines) is linear
    in the output gradient and reproducible to the atomics' reordering."""
    from fusionsense_amd.fused import render_fusionsense_fused
    params, cams = surface4
    g = torch.Generator().manual_seed(1)
    v = [{k: torch.randn(s, generator=g).to(dev) for k, s in (("rgb", (H4, W4, 3)), ("depth", (H4, W4, 1)),
                                                               ("normal", (H4, W4, 3)))} for _ in range(2)]

    def grads(weights):
        p = {k: t.clone().requires_grad_(True) for k, t in params.items()}
        out = render_fusionsense_fused(p, cams[5], sh_degree=3, device=dev)
        torch.autograd.backward([out[k] for k in ("rgb", "depth", "normal")],
                                [sum(w * vi[k] for w, vi in zip(weights, v)) for k in ("rgb", "depth", "normal")])
        return {k: t.grad for k, t in p.items()}

    g1, g2, g12, g12b = grads((1.0, 0.0)), grads((0.0, 1.0)), grads((2.0, 1.0)), grads((2.0, 1.0))
    for k in g12:
        assert bool(torch.isfinite(g12[k]).all())
        assert rel_err(g12[k], 2.0 * g1[k] + g2[k]) < 2e-3, k
        assert rel_err(g12b[k], g12[k]) < 5e-4, k
    assert float(g12["features_rest"].abs().max()) > 0 and float(g12["means"].abs().max()) > 0


def _bench_loss_oracle(tgt_cpu):
    from oracle import loss_ref

    def loss_of(out, to):  # the benchmark loss of SURVEY.md §8d (ops._TrainLoss): 0.8 L1 + 0.2 (1 - SSIM), 0.2 L1, 0.1 L1
        return (loss_ref.photometric_loss(out["rgb"], tgt_cpu["rgb"], None, 0.2)
                + 0.2 * (out["depth"] - tgt_cpu["depth"]).abs().mean()
                + 0.1 * (out["normal"] - tgt_cpu["normal"]).abs().mean())
    return loss_of


def _bench_route(dev, params, tgt, seen, **kw):
    """``route(cam, loss_of) -> (outputs, grads)``: SplatTrainer.train_step with the benchmark loss on ``tgt``, second
    visit of the view (binned with the occlusion cuts the first visit left)."""
    from fusionsense_amd.trainer import PARAM_ORDER

    def route(cam, _loss):
        tr, _ = _stats_trainer(dev, params, 1, **kw)
        tr.train_step(cam, tgt, optimizer_step=False)
        loss, out = tr.train_step(cam, tgt, optimizer_step=False)
        seen["cut_frames"], seen["loss"] = tr.cut_frames, float(loss)
        return out, {k: tr.slab.views[k] for k in PARAM_ORDER}
    return route


@pytest.mark.parametrize("walk", ["product", "one_wave_with_handoff"])
def test_config4_surface_scene_against_oracle_on_a_window(dev, surface4, walk):
    """All 6 M Gaussians of the surface scene against the CPU oracle on a 128x128 window of ring view 0 (the oracle needs
    ~40 s for it), two routes against one oracle frame: (a) the fused autograd node with a seeded weighted loss, (b) the
    route bench.py takes — SplatTrainer.train_step with the benchmark loss, second visit of the view (occlusion cuts on) —
    against the oracle's render + the oracle's SSIM / L1 loss.  Integer outputs, images, every parameter gradient; every
    mismatch must be an fp32 threshold case."""
    from fusionsense_amd import fused
    from helpers import OracleWindow, fused_node_route, shared_oracle_window
    if walk != "product":  # (the product's walk for a scene this dense IS the plain one-wave walk: force the hand-off on)
        fused.FWD_WALK.forced, fused.FWD_WALK.forced_walk = True, 1
        fused.FWD_WALK.handoff_records, fused.FWD_WALK.handoff_rel_len, fused.FWD_WALK.handoff_gate_len = 128, 0, 0
    params, cams = surface4
    cam = crop_camera(cams[0], CROP, shift=CROP_SHIFT)
    # (ONE oracle frame and one oracle backward per loss for both walks: the oracle's side does not depend on the walk)
    win = shared_oracle_window("config4_view0_window", lambda: OracleWindow({k: v.cpu() for k, v in params.items()}, cam))
    og, grads = fused_node_route(dev, params)(cam, win.loss_of)
    rep = win.check(og, grads, max_bad_radii=int(1e-4 * N4))
    assert rep["visible"] > 50_000 and rep["n_isects"] > 1024 * 64, rep  # (a dense window: the slab sorts ran)
    del og, grads
    tgt = _targets(dev, CROP, CROP, 21)
    seen = {}
    og, grads = _bench_route(dev, params, tgt, seen)(cam, None)
    rep = win.check(og, grads, max_bad_radii=int(1e-4 * N4), oracle_loss=_bench_loss_oracle({k: v.cpu() for k, v in tgt.items()}),
                    loss_key="bench_loss_targets_seed21")
    assert seen["cut_frames"] >= 1
    assert abs(seen["loss"] - rep["oracle_loss"]) <= 2e-5 * abs(rep["oracle_loss"]), (seen["loss"], rep["oracle_loss"])
    assert fused.FWD_WALK.error(dev) == 0


# ---------------------------------------------------------------------------------------------------------------
# config #5: 10 M Gaussians, half attribute storage, the same scene recipe
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def surface5(dev):
    from fusionsense_amd.trainer import HALF_GROUPS
    params = scenes.surface_dense_scene(N5, seed=0)
    rounded = {k: (v.half().float() if k in HALF_GROUPS else v) for k, v in params.items()}
    return params, rounded, _ring()


def test_config5_surface_scene_half_storage_bench_route(dev, surface5, fwd_walk):
    """10 M Gaussians, attributes read from half mirrors, the ring views at 1080p through SplatTrainer.train_step: lists
    sorted / consistent / equal to the radix chain, forward bit-reproducible and equal to the fp32 path on the rounded
    values up to a handful of threshold pixels, the cut frame equal to the uncut one, steps finite and allocation-free,
    mirrors == master.half()."""
    from fusionsense_amd.fused import render_fusionsense_fused
    from fusionsense_amd.trainer import HALF_GROUPS
    params, rounded, cams = surface5
    view = cams[2]
    tr, st = _stats_trainer(dev, params, 1, half_attributes=True)
    hm = tr.half_mirrors()
    assert sum(t.numel() * 2 for t in hm.values()) == N5 * 56 * 2  # 112 B of half attributes (+ 12 B fp32 means = 124)
    tgt = _targets(dev, H4, W4, 9)
    l0, o0 = tr.train_step(view, tgt, optimizer_step=False)
    img0 = {k: o0[k].clone() for k in ("rgb", "depth", "normal", "accumulation")}
    M, n_tile = _check_lists(dev, tr.params, o0["info"], W4, H4, min_pairs=8_000_000, half=hm)
    assert int(n_tile.max()) > 8192
    _check_last_ids(dev, o0, W4, H4)
    l1, o1 = tr.train_step(view, tgt, optimizer_step=False)
    assert tr.cut_frames >= 1
    for k in img0:
        assert torch.equal(o1[k], img0[k]) and bool(torch.isfinite(o1[k]).all()), k
    assert float((img0["accumulation"] > 0.99).float().mean()) > 0.5
    with torch.no_grad():
        rp = {k: v.to(dev) for k, v in rounded.items()}
        c = render_fusionsense_fused(rp, view, sh_degree=3, device=dev)
    for k in img0:
        # the half instantiation of the projecting count pass is a separately compiled inline of the same statements
        # (FMA contraction may differ): equal to rounding, a handful of pixels on an alpha >= 1/255 decision
        d = (img0[k] - c[k]).abs()
        assert float((d > 1e-4 * max(1.0, float(c[k].abs().max()))).float().mean()) < 1e-4, k
    assert abs(M - c["info"].payload.numel()) <= 1e-5 * c["info"].payload.numel()
    del rp, c
    tr.train_step(view, tgt)
    tr.train_step(view, tgt)
    torch.cuda.synchronize()
    n_alloc = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    for _ in range(3):
        loss, _ = tr.train_step(view, tgt)
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - n_alloc <= 1
    assert math.isfinite(float(loss))
    for k in HALF_GROUPS:
        assert torch.equal(tr.half_mirrors()[k], tr.params[k].data.half()), k
    # ... and a step of this configuration launches libfsgs kernels only: no torch (at::native) kernel, no fill, no copy
    # (the 594 us torch fills of profiles/r4_c5_kernel_stats.csv are the trainer's construction — zeroed
    # moments, slab and mirrors —, not its steps)
    from fusionsense_amd import fused
    from helpers import device_kernels_of
    # (THAT fill was the round-4 dispatch-order tuner's 640 MB scratch accumulator; the order is a rule since round 5)
    names = device_kernels_of(lambda: [tr.train_step(view, tgt) for _ in range(2)])
    assert names, "the profiler saw no device activity"
    theirs = ("at::", "rocprim", "hipcub", "Cijk_", "__amd_rocclr", "emcpy", "emset", "elementwise", "reduce_kernel")
    foreign = [n for n in names if any(t in n for t in theirs)]
    assert any("raster_bwd_live_kernel" in n for n in names) and any("gauss_sh_bwd_kernel" in n for n in names), names[:8]
    assert not foreign, sorted(set(foreign))


def test_config5_bench_step_against_oracle_on_a_window(dev, surface5):
    """All 10 M Gaussians, half storage, through SplatTrainer.train_step with the benchmark loss on a 128x128 window of a
    ring view, second visit (cuts on), the product's forward walk: against the CPU
    oracle fed the fp16-ROUNDED attributes (means stay fp32) — integer outputs, images, the gradient slab; every mismatch
    an fp32 threshold case."""
    from fusionsense_amd import fused
    from helpers import OracleWindow
    params, rounded, cams = surface5
    assert fused.FWD_WALK.choice(N5)[0] == 1  # the product's walk (dense scenes: one wave per quadrant, no hand-off)
    cam = crop_camera(cams[0], CROP, shift=CROP_SHIFT)
    tgt = _targets(dev, CROP, CROP, 33)
    seen = {}
    og, grads = _bench_route(dev, params, tgt, seen, half_attributes=True)(cam, None)
    win = OracleWindow(rounded, cam)
    rep = win.check(og, grads, max_bad_radii=int(1e-4 * N5), oracle_loss=_bench_loss_oracle({k: v.cpu() for k, v in tgt.items()}))
    assert seen["cut_frames"] >= 1 and rep["n_isects"] > 1024 * 64
    assert abs(seen["loss"] - rep["oracle_loss"]) <= 2e-5 * abs(rep["oracle_loss"]), (seen["loss"], rep["oracle_loss"])
