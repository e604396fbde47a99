"""GPU parity for the densify/prune rows: statistics kernel, row
compaction, split sampler and the full refinement_after against the CPU oracle."""
import numpy as np
import pytest
import torch

from fusionsense_amd import scenes
from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig, densify_masks
from fusionsense_amd.trainer import PARAM_ORDER, SplatTrainer
from oracle import splatfacto_ref

pytestmark = pytest.mark.gpu


def test_densify_stats_kernel(dev):
    from fusionsense_amd import ops
    g = torch.Generator().manual_seed(0)
    n = 5000
    radii = (torch.rand(n, generator=g) * 40 - 10).clamp(min=0).to(torch.int32)
    absgrad = torch.rand(1, n, 2, generator=g)
    state = {}
    for it in range(3):
        splatfacto_ref.after_train(state, radii, absgrad[0] * (it + 1), 720, 1280, it, 10000)
    xg = torch.zeros(n, device=dev)
    vc = torch.ones(n, device=dev)
    m2 = torch.zeros(n, device=dev)
    for it in range(3):
        ops.densify_stats_(radii.to(dev), (absgrad[0] * (it + 1)).to(dev), 1280, xg, vc, m2)
    assert torch.allclose(xg.cpu(), state["xys_grad_norm"], rtol=1e-6, atol=1e-7)
    assert torch.equal(vc.cpu(), state["vis_counts"])
    assert torch.allclose(m2.cpu(), state["max_2Dsize"], rtol=1e-6)


@pytest.mark.parametrize("n,row", [(0, 3), (1, 3), (2049, 1), (10007, 45), (4096, 4)])
def test_compact_rows_bit_exact(dev, n, row):
    from fusionsense_amd import ops
    g = torch.Generator().manual_seed(n)
    src = torch.randn(n, row, generator=g)
    keep = torch.rand(n, generator=g) > 0.4
    if n == 0:
        return
    keep8 = keep.to(torch.uint8).to(dev)
    pos = ops.mask_positions(keep8)
    out = ops.compact_rows(src.to(dev), keep8, pos, int(keep.sum()))
    assert torch.equal(out.cpu(), src[keep])


def test_split_samples_kernel(dev):
    from fusionsense_amd import ops
    from oracle.gsplat_ref import quat_to_rotmat
    g = torch.Generator().manual_seed(1)
    n, S, samps = 300, 40, 2
    means, quats, ls = torch.randn(n, 3, generator=g), torch.randn(n, 4, generator=g), torch.randn(n, 3, generator=g) - 3
    ids = torch.randperm(n, generator=g)[:S].sort().values
    z = torch.randn(samps * S, 3, generator=g)
    nm, nls = ops.split_samples(ids.to(dev), samps, means.to(dev), quats.to(dev), ls.to(dev), z.to(dev))
    q = quats[ids] / quats[ids].norm(dim=-1, keepdim=True)
    ref_m = torch.bmm(quat_to_rotmat(q.repeat(samps, 1)), (torch.exp(ls[ids].repeat(samps, 1)) * z)[..., None]).squeeze(-1) \
        + means[ids].repeat(samps, 1)
    ref_s = torch.log(torch.exp(ls[ids]) / 1.6).repeat(samps, 1)
    assert torch.allclose(nm.cpu(), ref_m, atol=1e-5)
    assert torch.allclose(nls.cpu(), ref_s, atol=1e-6)


def _trainer_with_state(dev, n, seed, step):
    params, _ = scenes.cube_scene(n, seed=seed)
    g = torch.Generator().manual_seed(seed)
    params["scales"] = torch.log(0.002 + 0.03 * torch.rand(n, 3, generator=g))
    params["opacities"] = 2 * torch.randn(n, 1, generator=g)
    st = DensifyStrategy(SplatfactoConfig(), num_train_data=9)
    tr = SplatTrainer(params, dev, strategy=st, seed=77)
    # give Adam non-trivial moments
    for k in PARAM_ORDER:
        tr.params[k].grad.copy_(torch.randn(tr.params[k].shape, generator=g).to(dev))
    for opt in tr.optimizers.values():
        opt.step()
    tr.step = step
    st.xys_grad_norm = (torch.rand(n, generator=g) * 0.01).to(dev)
    st.vis_counts = (1 + torch.randint(0, 5, (n,), generator=g).float()).to(dev)
    st.max_2Dsize = (torch.rand(n, generator=g) * 0.2).to(dev)
    st.last_size = (720, 1280)
    return tr, st


@pytest.mark.parametrize("step", [700, 3100, 3500, 10000])
def test_refinement_after_matches_oracle(dev, step):
    cfg = SplatfactoConfig()
    n = 3000
    tr, st = _trainer_with_state(dev, n, 3, step)
    params_cpu = {k: v.detach().cpu().clone() for k, v in tr.params.items()}
    adam_cpu = {k: {m: tr.optimizers[k].state[tr.params[k]][m].cpu().clone() for m in ("exp_avg", "exp_avg_sq")}
                for k in PARAM_ORDER}
    state_cpu = dict(xys_grad_norm=st.xys_grad_norm.cpu().clone(), vis_counts=st.vis_counts.cpu().clone(),
                     max_2Dsize=st.max_2Dsize.cpu().clone())
    splits, _ = densify_masks(cfg, step, state_cpu["xys_grad_norm"], state_cpu["vis_counts"],
                              state_cpu["max_2Dsize"], params_cpu["scales"], (720, 1280), None)
    g2 = torch.Generator(device=dev)
    g2.manual_seed(77)
    z = torch.randn((cfg.n_split_samples * int(splits.sum()), 3), device=dev, generator=g2).cpu()
    ref_p, ref_a, rep = splatfacto_ref.refinement_after(params_cpu, adam_cpu, state_cpu, cfg, step, 9, (720, 1280), z)

    st.refinement_after(tr, step)
    assert tr.params["means"].shape[0] == ref_p["means"].shape[0], (st.last_report, rep)
    for k in PARAM_ORDER:
        assert torch.allclose(tr.params[k].detach().cpu(), ref_p[k], atol=1e-5), k
        s = tr.optimizers[k].state[tr.params[k]]
        assert torch.equal(s["exp_avg"].cpu(), ref_a[k]["exp_avg"]), k
        assert torch.equal(s["exp_avg_sq"].cpu(), ref_a[k]["exp_avg_sq"]), k
        assert tr.params[k].grad.data_ptr() == tr.slab.views[k].data_ptr()
    assert st.xys_grad_norm is None and st.max_2Dsize is None
    if step in (700, 3500):
        assert rep["n_split"] > 0 and rep["n_dup"] > 0 and rep["n_deleted"] > 0
    # the model still trains after surgery
    cam = scenes.cube_scene(10)[1]
    out = tr.forward(cam)
    out["rgb"].mean().backward()
    for opt in tr.optimizers.values():
        opt.step()


def test_touch_and_hull_callbacks(dev):
    from fusionsense_amd import touch
    n = 2000
    tr, st = _trainer_with_state(dev, n, 5, 1000)
    means = tr.params["means"].detach().clone()
    g = torch.Generator().manual_seed(0)
    ax = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    ext = torch.tensor([0.2, 0.2, 0.1])
    c = torch.tensor([0.5, 0.0, 0.0])
    box = torch.stack([c + (ax * ((torch.tensor([(i >> 0) & 1, (i >> 1) & 1, (i >> 2) & 1]).float() * 2 - 1) * ext)[None]).sum(-1)
                       for i in range(8)])
    pts = c + 0.01 * torch.randn(50, 3, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(50, 3, generator=g), dim=-1)
    patches = [dict(points_xyz=pts, normals=nrm, bbox=box), dict(points_xyz=torch.zeros(0, 3), normals=torch.zeros(0, 3), bbox=box)]
    inside = touch.points_in_non_aabb(means.cpu(), box)
    low_alpha = (torch.sigmoid(tr.params["opacities"].detach().cpu()) < 0.1).squeeze(-1)
    expected_after_cull = n - int((inside | low_alpha).sum())
    added = st.add_touch_patch(tr, patches, gel_scale_factor=6.34e-4)
    assert added == 50
    assert tr.params["means"].shape[0] == expected_after_cull + 50
    assert int(st.add_mask.sum()) == 50 and st.add_mask[-50:].all()
    assert torch.allclose(tr.params["means"][-50:].detach().cpu(), pts, atol=1e-6)
    assert (tr.params["opacities"][-50:] == 1).all()
    # anchors survive touch pruning; everything else inside the box is gone
    tr.params["means"].data[:5] = c.to(dev)  # move 5 ordinary Gaussians into the box
    tr.step = 1100
    deleted = st.touch_pruning(tr, patches)
    assert deleted is not None and int(deleted[:5].sum()) == 5
    assert int(st.add_mask.sum()) == 50
    # hull pruning: points in the (0.005, 0.02] shell around the hull are culled
    hull = torch.zeros(1, 3)
    m = tr.params["means"]
    m.data[:3] = torch.tensor([[0.01, 0, 0], [0.001, 0, 0], [0.05, 0, 0]], device=dev)
    tr.params["opacities"].data[:3] = 5.0
    n_before = m.shape[0]
    d = st.hull_pruning(tr, hull, scale_factor=1.0)
    assert bool(d[0]) and not bool(d[1]) and not bool(d[2])
    assert tr.params["means"].shape[0] == n_before - int(d.sum())


def test_row_moving_callbacks_flush_a_deferred_feature_update(dev):
    """add_touch_patch is a BEFORE_TRAIN_ITERATION callback; in a data-parallel run the previous
    step's deferred feature update (sized for the old N, reading the old slab) is still pending when it runs.
    _rebuild must land it first.  Forced split step on one rank, both exchange formats, against the plain step."""
    from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig
    from fusionsense_amd.trainer import PARAM_ORDER, SplatTrainer
    params = scenes.lego_like_scene(6000, seed=5)
    cams = scenes.hemisphere_cameras(2, width=160, height=128, focal=180.0, seed=1)
    g = torch.Generator().manual_seed(0)
    tgts = [{"rgb": torch.rand(128, 160, 3, generator=g).to(dev), "depth": torch.rand(128, 160, 1, generator=g).to(dev),
             "normal": torch.rand(128, 160, 3, generator=g).to(dev)} for _ in cams]
    pts = 0.05 * torch.randn(40, 3, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(40, 3, generator=g), dim=-1)
    box = torch.tensor([[(i & 1) * 0.2 - 0.1, ((i >> 1) & 1) * 0.2 - 0.1, ((i >> 2) & 1) * 0.2 - 0.1] for i in range(8)])
    patches = [dict(points_xyz=pts, normals=nrm, bbox=box)]
    for factored in (True, False):
        trs = []
        for split in (True, False):
            st = DensifyStrategy(SplatfactoConfig(), num_train_data=2, stats_only=True)
            tr = SplatTrainer(params, dev, seed=0, strategy=st)
            tr.force_split_step = split
            tr.ride_adam = False  # (the plain step for split = False: nothing deferred)
            tr.factored_features = factored
            tr.train_step(cams[0], tgts[0])
            assert (tr._pending is not None) == split
            added = st.add_touch_patch(tr, patches, gel_scale_factor=6.34e-4)
            assert added == 40 and tr._pending is None
            tr.train_step(cams[1], tgts[1])
            tr.flush()
            trs.append(tr)
        a, b = trs
        assert a.num_gaussians() == b.num_gaussians()
        for k in PARAM_ORDER:
            d = (a.params[k].data - b.params[k].data).abs()
            assert float((d > 2e-5).float().mean()) < 2e-3 and float(d.mean()) < 2e-6, (factored, k)


@pytest.mark.parametrize("nq,np_", [(1, 1), (300, 1), (1000, 1025), (5000, 4097), (70, 20000)])
def test_nearest_point_kernel(dev, nq, np_):
    """fsgs_nearest_point (hull_pruning's cdist(...).min, add_touch_patch's 1-NN) against exhaustive fp64 search."""
    from fusionsense_amd import ops
    g = torch.Generator().manual_seed(nq * 7 + np_)
    q, p = torch.randn(nq, 3, generator=g), torch.randn(np_, 3, generator=g)
    if np_ > 10:
        p[7] = p[3]  # duplicated point: ties -> lowest index
        q[0] = p[3]
    d, i = ops.nearest_point(q.to(dev), p.to(dev), want_idx=True)
    ref = torch.cdist(q.double(), p.double())
    rd, ri = ref.min(dim=-1)
    assert torch.allclose(d.cpu().double(), rd, atol=1e-6)
    chosen = ref[torch.arange(nq), i.cpu()]
    assert torch.allclose(chosen, rd, atol=1e-6), "the returned index is a nearest point"
    if np_ > 10:
        assert int(i[0]) == 3
    # distances only (no index bookkeeping), and with a mask of the queries that matter: the same values, +inf elsewhere
    assert torch.equal(ops.nearest_point(q.to(dev), p.to(dev)), d)
    act = torch.rand(nq, generator=g) < 0.3
    act[: min(nq, 300)] = False  # (whole workgroups without an active query)
    da, ia = ops.nearest_point(q.to(dev), p.to(dev), want_idx=True, active=act.to(dev))
    assert torch.equal(da[act.to(dev)], d[act.to(dev)]) and torch.equal(ia[act.to(dev)], i[act.to(dev)])
    assert bool(torch.isinf(da[~act.to(dev)]).all()) and int(ia[~act.to(dev)].abs().sum()) == 0


# ---- the product's callbacks against goldens made by EXECUTING the reference's own methods ----------------------
import os  # noqa: E402

NAMES6 = ["means", "scales", "quats", "features_dc", "features_rest", "opacities"]


def _ref_model():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_model.npz"))


def _trainer_from_golden(dev, d, prefix, m_prefix, v_prefix=None, step=0, num_train_data=9):
    t = lambda k: torch.from_numpy(d[k])  # noqa: E731
    st = DensifyStrategy(SplatfactoConfig(), num_train_data=num_train_data)
    tr = SplatTrainer({k: t(f"{prefix}.{k}") for k in NAMES6}, dev, strategy=st, seed=0)
    for k in NAMES6:
        p = tr.params[k]
        m = t(f"{m_prefix}.{k}").to(dev)
        v = t(f"{v_prefix}.{k}").to(dev) if v_prefix else torch.zeros_like(m)
        tr.optimizers[k].state[p] = {"step": torch.tensor(1.0), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
    tr.step = step
    return tr, st


@pytest.mark.parametrize("step", [400, 700, 3100, 3500, 10000])
def test_refinement_after_matches_reference_execution(dev, step, monkeypatch):
    """DensifyStrategy.refinement_after (HIP row compaction, split sampler) against the reference's own
    refinement_after, executed in the authoring container (reference_model.npz): same split noise, same statistics —
    parameters and both Adam moments, row for row."""
    d = _ref_model()
    t = lambda k: torch.from_numpy(d[k])  # noqa: E731
    tr, st = _trainer_from_golden(dev, d, f"ra.{step}.pre", f"ra.{step}.pre_m", f"ra.{step}.pre_v", step)
    st.xys_grad_norm, st.vis_counts, st.max_2Dsize = (t(f"ra.stats.{k}").to(dev) for k in ("xys_grad_norm", "vis_counts", "max_2Dsize"))
    st.last_size = (720, 1280)
    n = tr.num_gaussians()
    if step == 3500:
        st.add_mask = torch.zeros(n, dtype=torch.bool, device=dev)
        st.add_mask[:50] = True
    noise = t(f"ra.{step}.randn")
    real_randn = torch.randn

    def fake_randn(*size, **kw):
        shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
        if shape == tuple(noise.shape):
            return noise.to(kw.get("device", "cpu"))
        kw.pop("generator", None)
        return real_randn(*size, **kw)

    monkeypatch.setattr(torch, "randn", fake_randn)
    st.refinement_after(tr, step)
    for k in NAMES6:
        ref = t(f"ra.{step}.post.{k}")
        assert tr.params[k].shape == ref.shape, (k, tr.params[k].shape, ref.shape, st.last_report)
        assert torch.allclose(tr.params[k].detach().cpu(), ref, atol=1e-5), k
        s_ = tr.optimizers[k].state[tr.params[k]]
        assert torch.equal(s_["exp_avg"].cpu(), t(f"ra.{step}.post_m.{k}")), k
        assert torch.equal(s_["exp_avg_sq"].cpu(), t(f"ra.{step}.post_v.{k}")), k
    am = d[f"ra.{step}.add_mask_after"]
    if am.size:
        assert np.array_equal(st.add_mask.cpu().numpy(), am)
    assert (st.xys_grad_norm is None) == bool(d[f"ra.{step}.stats_cleared"])


def test_touch_and_hull_callbacks_match_reference_execution(dev):
    """add_touch_patch, touch_pruning and hull_pruning against the reference's own methods executed on the same
    state (reference_model.npz): the appended anchor rows (positions, nearest-neighbour colours, gel scales,
    orientation from the contact normals), the culled rows, zero Adam moments for the new rows, add_mask."""
    d = _ref_model()
    t = lambda k: torch.from_numpy(d[k])  # noqa: E731
    tr, st = _trainer_from_golden(dev, d, "tp.in", "tp.in_m", None, 1000)
    patches = [{k: t(f"tp.patch{i}.{k}") for k in ("points_xyz", "points_rgb", "normals", "bbox")} for i in range(3)]
    hull = t("tp.hull")
    added = st.add_touch_patch(tr, patches, gel_scale_factor=6.34e-5)
    assert added == int(d["tp.add_mask"].sum())
    assert np.array_equal(st.add_mask.cpu().numpy(), d["tp.add_mask"])
    for k in NAMES6:
        ref = t(f"tp.added.{k}")
        assert tr.params[k].shape == ref.shape, k
        assert torch.allclose(tr.params[k].detach().cpu(), ref, atol=1e-6), (k, float((tr.params[k].detach().cpu() - ref).abs().max()))
        assert torch.equal(tr.optimizers[k].state[tr.params[k]]["exp_avg"].cpu(), t(f"tp.added_m.{k}")), k
    assert torch.allclose(st.touch_normals.cpu(), torch.cat([p["normals"] for p in patches]))
    tr.step = 1100
    tr.params["means"].data.copy_(t("tp.before_touch_prune.means").to(dev))
    st.touch_pruning(tr, patches)
    for k in NAMES6:
        assert torch.allclose(tr.params[k].detach().cpu(), t(f"tp.touch_pruned.{k}"), atol=1e-6), k
    assert np.array_equal(st.add_mask.cpu().numpy(), d["tp.add_mask_touch_pruned"])
    st.hull_pruning(tr, hull, scale_factor=1.3)
    for k in NAMES6:
        assert tr.params[k].shape == t(f"tp.hull_pruned.{k}").shape, k
        assert torch.allclose(tr.params[k].detach().cpu(), t(f"tp.hull_pruned.{k}"), atol=1e-6), k
    assert np.array_equal(st.add_mask.cpu().numpy(), d["tp.add_mask_hull_pruned"])
    # ... and the two prunings as maybe_refine runs them since round 5 — ONE cull over the union of the two tests — leave
    # the same rows, moments and anchors, bit for bit
    tr2, st2 = _trainer_from_golden(dev, d, "tp.in", "tp.in_m", None, 1000)
    st2.set_metadata(touch_patches=patches, gel_scale_factor=6.34e-5, add_touch_at=1000, visual_hull=hull, scale_factor=1.3)
    st2.stage(dev)
    st2.add_touch_patch(tr2, st2.touch_patches, gel_scale_factor=6.34e-5)
    tr2.step = 1100
    tr2.params["means"].data.copy_(t("tp.before_touch_prune.means").to(dev))
    deleted = st2.hull_and_touch_pruning(tr2)
    assert deleted is not None and int(deleted.sum()) > 0
    for k in NAMES6:
        assert torch.equal(tr2.params[k].detach(), tr.params[k].detach()), k
        for mom in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(tr2.optimizers[k].state[tr2.params[k]][mom], tr.optimizers[k].state[tr.params[k]][mom]), (k, mom)
    assert torch.equal(st2.add_mask, st.add_mask)


def test_points_in_boxes_kernel(dev):
    """fsgs_points_in_boxes (one launch for all patches) against the per-patch points_in_non_aabb loop of the reference
    (dn_model.py:1996-2034, mirrored in touch.py and pinned there by reference_helpers.npz): the same mask except for
    points within rounding of a face (the kernel's dot product is an fma chain, torch's a matrix product)."""
    from fusionsense_amd import ops, touch
    g = torch.Generator().manual_seed(11)
    patches = []
    for b in range(4):
        ax = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
        ext = 0.05 + 0.3 * torch.rand(3, generator=g)
        c = 0.4 * torch.randn(3, generator=g)
        box = torch.stack([c + (ax * ((torch.tensor([(i >> 0) & 1, (i >> 1) & 1, (i >> 2) & 1]).float() * 2 - 1) * ext)[None]).sum(-1)
                           for i in range(8)])
        patches.append(dict(points_xyz=torch.zeros(0 if b == 2 else 5, 3), bbox=box))  # (an empty patch has no box)
    pts = 0.6 * torch.randn(20001, 3, generator=g)
    ref = torch.zeros(pts.shape[0], dtype=torch.bool)
    margin = torch.full((pts.shape[0],), float("inf"))
    for patch in patches:
        if patch["points_xyz"].shape[0] == 0:
            continue
        ref |= touch.points_in_non_aabb(pts, patch["bbox"])
        bv = patch["bbox"]
        axes = torch.nn.functional.normalize(torch.stack([bv[1] - bv[0], bv[2] - bv[0], bv[4] - bv[0]]), dim=1)
        rel_p, rel_c = (pts - bv.mean(0)) @ axes.T, (bv - bv.mean(0)) @ axes.T
        margin = torch.minimum(margin, torch.minimum((rel_p - rel_c.min(0).values).abs(), (rel_p - rel_c.max(0).values).abs()).min(-1).values)
    frames = touch.box_frames(patches, dev)
    assert frames.shape == (3, 18)
    got = ops.points_in_boxes(pts.to(dev), frames).cpu()
    assert got.dtype == torch.bool and 0.02 < float(ref.float().mean()) < 0.9
    differ = got != ref
    assert not bool((differ & (margin > 1e-5)).any()), int(differ.sum())
    assert torch.equal(touch.touch_aabb_mask(pts.to(dev), patches).cpu(), got)
    assert ops.points_in_boxes(pts.to(dev), frames[:0]).sum() == 0  # no boxes: nothing is inside


# ---- the two launches of a refinement against their oracle (oracle/refine_ref.py, pinned against the reference's own
# refinement_after by tests/test_host_logic.py::test_refinement_host_logic_with_the_row_oracle_matches_reference_execution)
def _refine_case(n, seed, sh_rest=15):
    g = torch.Generator().manual_seed(seed)
    cols = dict(means=torch.randn(n, 3, generator=g), scales=torch.log(0.002 + 0.03 * torch.rand(n, 3, generator=g)),
                quats=torch.randn(n, 4, generator=g), features_dc=torch.randn(n, 3, generator=g),
                features_rest=torch.randn(n, sh_rest, 3, generator=g), opacities=2 * torch.randn(n, 1, generator=g))
    stats = dict(xys_grad_norm=torch.rand(n, generator=g) * 0.01, vis_counts=1 + torch.randint(0, 5, (n,), generator=g).float(),
                 max_2Dsize=torch.rand(n, generator=g) * 0.2)
    add = torch.rand(n, generator=g) < 0.02
    return cols, stats, add


def _mark_both(dev, rules, cols, stats, add, **kw):
    from fusionsense_amd import ops
    from oracle import refine_ref
    to = lambda x: None if x is None else x.to(dev)  # noqa: E731
    dense = bool(rules.densify)
    ref = refine_ref.mark(rules, cols["scales"], cols["opacities"], stats["xys_grad_norm"] if dense else None,
                          stats["vis_counts"] if dense else None, stats["max_2Dsize"], cols["means"], add, **kw)
    got = ops.refine_mark(rules, to(cols["scales"]), to(cols["opacities"]), to(stats["xys_grad_norm"]) if dense else None,
                          to(stats["vis_counts"]) if dense else None, to(stats["max_2Dsize"]), to(cols["means"]), to(add),
                          **{k: to(v) for k, v in kw.items()})
    return ref, got


@pytest.mark.parametrize("n,step,densify", [(70001, 700, True), (70001, 3500, True), (4096, 10000, False), (257, 16000, True),
                                            (1, 700, True), (0, 700, True)])
def test_refine_mark_matches_its_oracle(dev, n, step, densify):
    """fsgs_refine_mark: the flag byte of every row and the five totals, over the schedule's regimes (screen-size tests on
    / off, too-big test on / off, cull only) — equal to the torch statement except where a row sits within rounding of a
    threshold (expf / logf of the device against the host's)."""
    cols, stats, add = _refine_case(n, step + n)
    st = DensifyStrategy(SplatfactoConfig(), num_train_data=9)
    st.last_size = (720, 1280)
    rules = st._rules(step, densify)
    extra = torch.rand(n, generator=torch.Generator().manual_seed(1)) < 0.05
    ref, got = _mark_both(dev, rules, cols, stats, add, extra_cull=extra)
    differ = int((got.flags[:n].cpu() != ref.flags).sum())
    assert differ <= 2, differ
    if differ == 0:
        for k in ("n_keep", "n_split", "n_dup", "n_split_kept", "n_dup_kept", "n_samples"):
            assert getattr(got, k) == getattr(ref, k), k
        assert torch.equal(got.deleted().cpu(), ref.deleted())
    if n > 1000 and densify:
        assert ref.n_split > 0 and ref.n_dup > 0 and ref.n_split_kept < ref.n_split + 1 and ref.n_keep < n
        both = (ref.flags & 6) == 6
        assert step > 4000 or int(both.sum()) > 0  # the in-place shrink quirk: split parents that are also duplicated
    # the totals are the flags' own counts, whatever the rounding did
    f = got.flags[:n].cpu()
    assert got.n_keep == int((f & 1).ne(0).sum()) and got.n_split == int((f & 2).ne(0).sum())
    assert got.n_dup == int((f & 4).ne(0).sum()) and got.n_split_kept == int(((f & 10) == 10).sum())
    assert got.n_dup_kept == int(((f & 20) == 20).sum())


def test_refine_mark_pruning_tests_ride_along(dev):
    """hull_pruning's shell test (on fsgs_nearest_point_words' words) and touch_pruning's box test inside the mark: the
    same rows as the torch statement, anchors exempt; words equal to an exhaustive search for the close rows."""
    from fusionsense_amd import ops, touch
    from oracle import refine_ref
    n = 30011
    cols, stats, add = _refine_case(n, 5)
    cols["means"] = cols["means"] * 0.3
    g = torch.Generator().manual_seed(9)
    hull = 0.25 * torch.nn.functional.normalize(torch.randn(3000, 3, generator=g), dim=-1) + 0.01
    center = [float(x) for x in hull.mean(0).tolist()]
    s = 1.3
    words = ops.nearest_point_words(cols["means"].to(dev), hull.to(dev), center, 0.2 * s)
    ref_words = refine_ref.hull_distance_words(cols["means"], hull, center, 0.2 * s)
    close = ref_words != -1
    assert 0.1 < float(close.float().mean()) < 0.95
    assert torch.equal((words.cpu() == -1), ~close)
    d_got = (words.cpu() >> 32).to(torch.int32).view(torch.float32)[close]
    d_ref = (ref_words >> 32).to(torch.int32).view(torch.float32)[close]
    assert torch.allclose(d_got, d_ref, rtol=1e-5, atol=1e-9)
    boxes = []
    for b in range(3):
        ax = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
        ext = 0.05 + 0.1 * torch.rand(3, generator=g)
        c = 0.2 * torch.randn(3, generator=g)
        boxes.append(dict(points_xyz=torch.zeros(4, 3), bbox=torch.stack(
            [c + (ax * ((torch.tensor([(i >> 0) & 1, (i >> 1) & 1, (i >> 2) & 1]).float() * 2 - 1) * ext)[None]).sum(-1)
             for i in range(8)])))
    frames = touch.box_frames(boxes)
    st = DensifyStrategy(SplatfactoConfig(), num_train_data=9)
    rules = st._rules(1100, False)
    rules.hull_center[0], rules.hull_center[1], rules.hull_center[2] = center
    rules.hull_close, rules.hull_lo, rules.hull_hi = 0.2 * s, 0.005 * s, 0.02 * s
    stats["max_2Dsize"] = None
    ref = refine_ref.mark(rules, cols["scales"], cols["opacities"], None, None, None, cols["means"], add,
                          hull_words=words.cpu(), boxes=frames)
    got = ops.refine_mark(rules, cols["scales"].to(dev), cols["opacities"].to(dev), means=cols["means"].to(dev),
                          add_mask=add.to(dev), hull_words=words, boxes=frames.to(dev))
    plain = refine_ref.mark(st._rules(1100, False), cols["scales"], cols["opacities"])
    assert ref.n_keep < plain.n_keep - 100, "the pruning tests cull rows of their own"
    differ = got.flags[:n].cpu() != ref.flags
    assert int(differ.sum()) <= 3, int(differ.sum())  # (rows within rounding of a face / of the shell)
    assert not bool(((got.flags[:n].cpu() & 1) == 0)[add & ((plain.flags & 1) != 0)].any()), "anchors pass both tests"
    # ... and with the search riding in the mark (the hull's points by grid cell, touch.hull_grid): the exhaustive
    # search's decisions, bit for bit — a row's nearest hull point matters only within hull_hi, i.e. within the 27 cells
    flags_words = got.flags[:n].clone()
    grid = touch.hull_grid(hull.to(dev), 0.02 * s)
    assert grid is not None and grid["cells"].numel() == grid["dims"][0] * grid["dims"][1] * grid["dims"][2] + 1
    assert int(grid["cells"][-1]) == hull.shape[0]
    got2 = ops.refine_mark(rules, cols["scales"].to(dev), cols["opacities"].to(dev), means=cols["means"].to(dev),
                           add_mask=add.to(dev), boxes=frames.to(dev), hull_grid=grid)
    assert torch.equal(got2.flags[:n], flags_words) and got2.n_keep == got.n_keep
    assert touch.hull_grid(hull.to(dev) * 1e4, 0.02 * s) is None  # (too many cells: the exhaustive search stays)


@pytest.mark.parametrize("n,step,samps,rest,n_append", [(70001, 700, 2, 15, 0), (5000, 3500, 3, 0, 0), (3001, 10000, 2, 15, 77),
                                                        (300, 700, 2, 3, 1), (0, 700, 2, 15, 5), (1, 16000, 2, 15, 0)])
def test_refine_move_matches_its_oracle(dev, n, step, samps, rest, n_append):
    """fsgs_refine_move from the device's own flags: kept rows, children sample by sample (sampled means, shrunk scales),
    duplicates (shrunk where the parent was split too), appended rows, zero moments for every new row, the anchor mask —
    copied columns bit for bit, computed ones to fp32 rounding."""
    from fusionsense_amd import ops
    from oracle import refine_ref
    cols, stats, add = _refine_case(n, step + n, rest)
    cfg = SplatfactoConfig()
    cfg.n_split_samples = samps
    st = DensifyStrategy(cfg, num_train_data=9)
    st.last_size = (720, 1280)
    densify = step < 15000
    rules = st._rules(step, densify)
    to = lambda x: None if x is None else x.to(dev)  # noqa: E731
    got = ops.refine_mark(rules, to(cols["scales"]), to(cols["opacities"]), to(stats["xys_grad_norm"]) if densify else None,
                          to(stats["vis_counts"]) if densify else None, to(stats["max_2Dsize"]), add_mask=to(add))
    ref = refine_ref.Marks()
    ref.n, ref.flags = n, got.flags[:n].cpu()
    for k in ("n_keep", "n_split", "n_dup", "n_split_kept", "n_dup_kept", "n_samples"):
        setattr(ref, k, getattr(got, k))
    g = torch.Generator().manual_seed(3)
    randn = torch.randn(max(got.n_samples * got.n_split, 1), 3, generator=g)
    n_out = got.n_keep + got.n_samples * got.n_split_kept + got.n_dup_kept + n_append
    kinds = dict(means=ops.COL_MEANS, scales=ops.COL_SCALES)
    app = {k: torch.randn((n_append,) + tuple(v.shape[1:]), generator=g) for k, v in cols.items()} if n_append else None
    job_d, job_h, outs = [], [], []
    for k, v in cols.items():
        for kind in (kinds.get(k, ops.COL_COPY), ops.COL_ZERO):  # the parameter, then a "moment" of the same rows
            dd = torch.full((n_out,) + tuple(v.shape[1:]), float("nan"), device=dev)
            dh = torch.empty((n_out,) + tuple(v.shape[1:]))
            a = app[k] if (app and kind != ops.COL_ZERO) else None
            job_d.append((v.to(dev), dd, a, kind))
            job_h.append((v, dh, a, kind))
            outs.append((k, kind, dd, dh))
    mo_d = torch.empty(n_out, dtype=torch.bool, device=dev)
    mo_h = torch.empty(n_out, dtype=torch.bool)
    ops.refine_move(got, job_d, n_append, to(randn), to(cols["quats"]), to(cols["scales"]), to(add), mo_d, 1)
    refine_ref.move(ref, job_h, n_append, randn, cols["quats"], cols["scales"], add, mo_h, 1)
    for k, kind, dd, dh in outs:
        if dd.numel() == 0:
            continue
        assert not bool(torch.isnan(dd).any()), (k, kind, "every row of the output is written")
        if kind in (ops.COL_COPY, ops.COL_ZERO):
            assert torch.equal(dd.cpu(), dh), (k, kind)
        else:
            assert torch.allclose(dd.cpu(), dh, rtol=1e-5, atol=1e-6), (k, kind, float((dd.cpu() - dh).abs().max()))
            assert torch.equal(dd.cpu()[:got.n_keep], dh[:got.n_keep])
    assert torch.equal(mo_d.cpu(), mo_h)
    if n > 1000 and densify:
        assert got.n_split_kept > 0 and got.n_dup_kept > 0


def test_refinement_is_two_library_launches_and_the_noise(dev):
    """What VERDICT r5 asked of a-13: one refinement_after = fsgs_refine_mark, torch.randn (the split noise, from the
    trainer's generator), fsgs_refine_move — and the gradient slab's memset; no other device work."""
    from helpers import device_kernels_of
    tr, st = _trainer_with_state(dev, 20000, 3, 700)
    st.refinement_after(tr, 700)  # (first call: scratch buffers, module loads)
    tr2, st2 = _trainer_with_state(dev, 20000, 3, 700)
    names = device_kernels_of(lambda: st2.refinement_after(tr2, 700))
    assert st2.last_report["n_split"] > 0 and st2.last_report["n_dup"] > 0
    ours = [n for n in names if "refine_" in n]
    assert len(ours) == 2 and "refine_mark_kernel" in ours[0] and "refine_move_kernel" in ours[1], names
    others = [n for n in names if "refine_" not in n]
    assert len(others) <= 3, others  # randn, the new slab's fill (+ a copy the profiler may show for the mapped totals)


def test_cull_moves_the_statistics_and_append_keeps_every_row(dev):
    """cull_gaussians between two refinements compacts the after_train statistics with the rows (a pruning callback must
    not shift them against the Gaussians they describe); append_gaussians keeps every row where it is, adds the new rows
    with zero moments and leaves the anchor mask's old entries alone."""
    tr, st = _trainer_with_state(dev, 5000, 11, 1100)
    before = {k: tr.params[k].detach().clone() for k in PARAM_ORDER}
    mom = {k: tr.optimizers[k].state[tr.params[k]]["exp_avg"].clone() for k in PARAM_ORDER}
    stats = {k: getattr(st, k).clone() for k in ("xys_grad_norm", "vis_counts", "max_2Dsize")}
    st.add_mask = torch.zeros(5000, dtype=torch.bool, device=dev)
    st.add_mask[::7] = True
    g = torch.Generator().manual_seed(2)
    extra = (torch.rand(5000, generator=g) < 0.3).to(dev)
    deleted = st.cull_gaussians(tr, extra)
    keep = ~deleted
    assert bool((deleted | ~extra).all()) and 0 < int(keep.sum()) < 5000 - int(extra.sum()) + 1
    n1 = int(keep.sum())
    assert tr.num_gaussians() == n1
    for k in PARAM_ORDER:
        assert torch.equal(tr.params[k].detach(), before[k][keep]), k
        assert torch.equal(tr.optimizers[k].state[tr.params[k]]["exp_avg"], mom[k][keep]), k
    for k, v in stats.items():
        assert torch.equal(getattr(st, k), v[keep]), k
    assert torch.equal(st.add_mask, (torch.arange(5000, device=dev) % 7 == 0)[keep])
    # append: 33 new rows behind the survivors
    rows = {k: torch.randn((33,) + tuple(before[k].shape[1:]), generator=g) for k in PARAM_ORDER}
    kept = {k: tr.params[k].detach().clone() for k in PARAM_ORDER}
    mask_before = st.add_mask.clone()
    st.append_gaussians(tr, rows)
    assert tr.num_gaussians() == n1 + 33
    for k in PARAM_ORDER:
        assert torch.equal(tr.params[k].detach()[:n1], kept[k]), k
        assert torch.equal(tr.params[k].detach()[n1:].cpu(), rows[k]), k
        s_ = tr.optimizers[k].state[tr.params[k]]
        assert torch.equal(s_["exp_avg"][:n1], mom[k][keep]) and not bool(s_["exp_avg"][n1:].any()), k
        assert not bool(s_["exp_avg_sq"][n1:].any()), k
    assert torch.equal(st.add_mask[:n1], mask_before) and not bool(st.add_mask[n1:].any())
