"""The trainer's step as ONE library call (fsgs_step_run; fusionsense_amd/express.py) against the launch-by-launch route
it replaces (fused.fused_step_forward_backward): same launches, same arguments — so the forward's outputs are the same
BITS and the update is the same up to the order of the backward's float atomics.  Reference: get_outputs of
/root/reference/dn_splatter/dn_model.py:469-671 + nerfstudio's after_train + Adam, as for the per-op route."""
import copy
import math

import pytest
import torch

from fusionsense_amd import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _scene(dev, n=70_000, w=304, h=208, views=3, seed=21):
    params = scenes.lego_like_scene(n, seed=seed)
    cams = scenes.hemisphere_cameras(views, width=w, height=h, focal=400.0, seed=seed)
    g = torch.Generator().manual_seed(seed)
    tgts = [{"rgb": torch.rand(h, w, 3, generator=g).to(dev), "depth": torch.rand(h, w, 1, generator=g).to(dev) * 4,
             "normal": torch.rand(h, w, 3, generator=g).to(dev)} for _ in cams]
    return params, cams, tgts


def _snapshot(tr):
    """Everything a step changes, copied: parameters, moments, counters, the views' bucket room, the capacity estimates."""
    from fusionsense_amd.trainer import PARAM_ORDER
    snap = {"p": {k: tr._params[k].detach().clone() for k in PARAM_ORDER}, "m": {}, "v": {}, "t": {}}
    for k in PARAM_ORDER:
        st = tr.optimizers[k].state[tr._params[k]]
        snap["m"][k], snap["v"][k], snap["t"][k] = st["exp_avg"].clone(), st["exp_avg_sq"].clone(), st["step"].clone()
    snap["step"], snap["adam_steps"] = tr.step, getattr(tr, "adam_steps", 0)
    snap["hist"] = {k: {"base": v["base"].clone(), "n_live": v["n_live"]} for k, v in tr._bin_hist.items()}
    snap["caps"] = copy.deepcopy(tr._live_caps.table)
    return snap


def _restore(tr, snap):
    from fusionsense_amd.trainer import PARAM_ORDER
    with torch.no_grad():
        for k in PARAM_ORDER:
            tr._params[k].copy_(snap["p"][k])  # (in place: the argument blocks keep their pointers)
            st = tr.optimizers[k].state[tr._params[k]]
            st["exp_avg"].copy_(snap["m"][k])
            st["exp_avg_sq"].copy_(snap["v"][k])
            st["step"].copy_(snap["t"][k])
    tr.step, tr.adam_steps = snap["step"], snap["adam_steps"]
    tr._bin_hist.clear()
    for k, v in snap["hist"].items():
        tr._bin_hist[k] = {"base": v["base"].clone(), "n_live": v["n_live"]}
    tr._live_caps.table.clear()
    tr._live_caps.table.update(copy.deepcopy(snap["caps"]))
    torch.cuda.synchronize()


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_express_step_equals_the_per_op_step(dev):
    """From ONE state, the same step through both routes: images, losses, lists, offsets, projection results — equal bit
    for bit; parameters and moments after the step equal within the reordering of the backward's float atomics; the
    step counters, the view's bucket room and the binning route's bookkeeping are the same."""
    from fusionsense_amd.trainer import PARAM_ORDER, SplatTrainer
    params, cams, tgts = _scene(dev)
    tr = SplatTrainer(params, dev, sh_degree=3, seed=3)
    for it in range(6):  # first visits (two-pass route), one revisit each: every view has bucket room now
        tr.train_step(cams[it % 3], tgts[it % 3])
    assert tr.express_steps == 3, "revisited views take the one-call route"
    snap = _snapshot(tr)
    res = {}
    for route in ("per_op", "express", "per_op2"):
        _restore(tr, snap)
        tr.express = route == "express"
        before = tr.express_steps
        loss, out = tr.train_step(cams[1], tgts[1])
        torch.cuda.synchronize()
        assert (tr.express_steps - before) == (1 if route == "express" else 0)
        info = out["info"]
        assert info.bin_route == "fill" and not tr.last_step_grads_valid
        res[route] = dict(loss=float(loss), rgb=out["rgb"].clone(), depth=out["depth"].clone(), normal=out["normal"].clone(),
                          acc=out["accumulation"].clone(), radii=out["radii"].clone(), n_live=int(info.n_live),
                          payload=info.payload[:int(info.n_live)].clone(), offsets=info.isect_offsets.clone(),
                          means2d=info.means2d.clone(), conics=info.conics.clone(), tpg=info.tiles_per_gauss.clone(),
                          last_ids=info.last_ids.clone(), normals_world=out["normals_world"].clone(),
                          absgrad=info.absgrad.clone(),
                          p={k: tr._params[k].detach().clone() for k in PARAM_ORDER},
                          m={k: tr.optimizers[k].state[tr._params[k]]["exp_avg"].clone() for k in PARAM_ORDER},
                          hist=tr._bin_hist[tr._view_key(cams[1])]["base"].clone(), step=tr.step, adam=tr.adam_steps,
                          t=float(tr.optimizers["means"].state[tr._params["means"]]["step"]))
    a, b, c = res["per_op"], res["express"], res["per_op2"]
    for k in ("rgb", "depth", "normal", "acc", "radii", "payload", "offsets", "means2d", "conics", "tpg", "last_ids",
              "normals_world", "hist"):
        assert torch.equal(a[k], b[k]), k
    assert a["loss"] == b["loss"] and a["n_live"] == b["n_live"]
    assert (a["step"], a["adam"], a["t"]) == (b["step"], b["adam"], b["t"])
    # the update: as close to the per-op route's as two runs of the per-op route are to each other (float atomics)
    # the update: the first moments are LINEAR in the step's gradient — equal within the reordering of the backward's
    # float atomics; the parameters themselves within 1e-3 of the largest update (Adam's m / sqrt(v) amplifies the
    # atomics' noise on elements whose gradient nearly cancels; c, a second per-op run, shows the same spread)
    for k in PARAM_ORDER:
        assert _rel(b["m"][k], a["m"][k]) <= 1e-4, k
        spread = _rel(c["p"][k] - snap["p"][k], a["p"][k] - snap["p"][k])
        assert _rel(b["p"][k] - snap["p"][k], a["p"][k] - snap["p"][k]) <= max(1e-3, 20 * spread), (k, spread)
    assert _rel(b["absgrad"], a["absgrad"]) <= 1e-4


def test_express_get_loss_dict_step_equals_the_per_op_step(dev):
    """The same comparison with a FrameBatch as supervision: get_loss_dict's terms (dn_model.py:673-925: masked L1 + SSIM,
    EdgeAwareLogL1 + TV on the depth, L1 + TV on the normals, the min-scale term) through fsgs_step_run's loss_kind 1
    against ops._FusionLoss launch by launch — loss and images bit for bit, the update within the atomics' noise."""
    from fusionsense_amd.losses import LossConfig, prepare_batch
    from fusionsense_amd.trainer import PARAM_ORDER, SplatTrainer
    params, cams, _ = _scene(dev, n=50_000, seed=33)
    g = torch.Generator().manual_seed(33)
    h, w = cams[0].height, cams[0].width
    cfg = LossConfig()
    fbs = [prepare_batch({"image": torch.rand(h, w, 3, generator=g), "sensor_depth": torch.rand(h, w, 1, generator=g) * 3 + 0.5,
                          "normal": torch.rand(h, w, 3, generator=g), "mask": torch.rand(h, w, 1, generator=g) > 0.3}, cfg, dev)
           for _ in cams]
    # touch anchors (dn_model.py:535-541, 893-902): the last 200 rows are frozen, with contact normals for the MSE term
    from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig
    N = params["means"].shape[0]
    st = DensifyStrategy(SplatfactoConfig(), num_train_data=3, stats_only=True)
    st.add_mask = torch.zeros(N, dtype=torch.bool, device=dev)
    st.add_mask[-200:] = True
    st.touch_normals = torch.nn.functional.normalize(torch.randn(200, 3, generator=g), dim=-1).to(dev)
    tr = SplatTrainer(params, dev, sh_degree=3, seed=3, loss_cfg=cfg, strategy=st)
    for it in range(6):
        tr.train_step(cams[it % 3], fbs[it % 3])
    assert tr.express_steps == 3
    snap = _snapshot(tr)
    res = {}
    for route in ("per_op", "express", "per_op2"):
        _restore(tr, snap)
        tr.express = route == "express"
        before = tr.express_steps
        loss, out = tr.train_step(cams[0], fbs[0])
        torch.cuda.synchronize()
        assert (tr.express_steps - before) == (1 if route == "express" else 0)
        res[route] = dict(loss=float(loss), rgb=out["rgb"].clone(), depth=out["depth"].clone(), normal=out["normal"].clone(),
                          p={k: tr._params[k].detach().clone() for k in PARAM_ORDER},
                          m={k: tr.optimizers[k].state[tr._params[k]]["exp_avg"].clone() for k in PARAM_ORDER})
    a, b, c = res["per_op"], res["express"], res["per_op2"]
    assert a["loss"] == b["loss"] and math.isfinite(a["loss"])
    for k in ("rgb", "depth", "normal"):
        assert torch.equal(a[k], b[k]), k
    for k in ("means", "opacities"):  # the anchors did not move on either route (their log-scales do: the min-scale
        # term reaches them directly, dn_model.py:817-819)
        assert torch.equal(b["p"][k][-200:], snap["p"][k][-200:]) and torch.equal(a["p"][k][-200:], snap["p"][k][-200:]), k
    assert not torch.equal(b["p"]["scales"][-200:], snap["p"]["scales"][-200:])
    for k in PARAM_ORDER:  # (tolerances as in test_express_step_equals_the_per_op_step)
        assert _rel(b["m"][k], a["m"][k]) <= 1e-4, k
        spread = _rel(c["p"][k] - snap["p"][k], a["p"][k] - snap["p"][k])
        assert _rel(b["p"][k] - snap["p"][k], a["p"][k] - snap["p"][k]) <= max(1e-3, 20 * spread), (k, spread)


def test_express_step_without_the_optimizer_writes_the_gradients(dev):
    """optimizer_step=False through the one-call route (adam.n_groups = 0: fsgs_gauss_sh_bwd writes the gradients into the
    trainer's slab) against the launch-by-launch route: same images and lists bit for bit, gradients equal within the
    atomics' reordering, parameters untouched — for both losses (the FrameBatch one adds the min-scale launch)."""
    from fusionsense_amd.losses import LossConfig, prepare_batch
    from fusionsense_amd.trainer import PARAM_ORDER, SplatTrainer
    params, cams, tgts = _scene(dev, n=40_000, seed=17)
    g = torch.Generator().manual_seed(17)
    h, w = cams[0].height, cams[0].width
    cfg = LossConfig()
    fbs = [prepare_batch({"image": torch.rand(h, w, 3, generator=g), "sensor_depth": torch.rand(h, w, 1, generator=g) * 3 + 0.5,
                          "normal": torch.rand(h, w, 3, generator=g), "mask": torch.rand(h, w, 1, generator=g) > 0.3}, cfg, dev)
           for _ in cams]
    for sup in (tgts, fbs):
        tr = SplatTrainer(params, dev, sh_degree=3, seed=3, loss_cfg=cfg)
        for it in range(3):
            tr.train_step(cams[it], sup[it], optimizer_step=False)  # first visits: two-pass route
        before = {k: tr._params[k].detach().clone() for k in PARAM_ORDER}
        res = {}
        for route in ("per_op", "express"):
            tr.express = route == "express"
            x0 = tr.express_steps
            loss, out = tr.train_step(cams[1], sup[1], optimizer_step=False)
            torch.cuda.synchronize()
            assert tr.express_steps - x0 == (1 if route == "express" else 0) and tr.last_step_grads_valid
            res[route] = dict(loss=float(loss), rgb=out["rgb"].clone(), payload=out["info"].payload.clone(),
                              g={k: tr.gradients()[k].clone() for k in PARAM_ORDER},
                              pg={k: tr._params[k].grad.clone() for k in PARAM_ORDER})
        a, b = res["per_op"], res["express"]
        assert a["loss"] == b["loss"] and torch.equal(a["rgb"], b["rgb"]) and torch.equal(a["payload"], b["payload"])
        for k in PARAM_ORDER:
            assert torch.equal(tr._params[k].detach(), before[k]), k
            assert float(a["g"][k].abs().max()) > 0 and _rel(b["g"][k], a["g"][k]) <= 2e-4, (k, _rel(b["g"][k], a["g"][k]))
            assert torch.equal(b["pg"][k], b["g"][k])


def test_express_outputs_survive_the_next_step_and_training_converges_alike(dev):
    """Losses and images are the caller's to keep (fresh tensors every step: a list of losses read at the end holds every
    step's own value), a frame's side outputs stay valid until the next-but-one step, and a short training run through
    the one-call route follows the per-op route's losses (first steps: identical parameters -> identical losses; later
    ones within what the atomics' noise, fed back through Adam, leaves)."""
    from fusionsense_amd.trainer import SplatTrainer
    params, cams, tgts = _scene(dev, n=50_000, seed=5)
    finals = {}
    for express in (True, False):
        tr = SplatTrainer(params, dev, sh_degree=3, seed=3)
        tr.express = express
        kept, images = None, []
        losses = []
        for it in range(15):
            loss, out = tr.train_step(cams[it % 3], tgts[it % 3])
            if kept is not None:
                assert torch.equal(kept[0], kept[1]), "the previous step's side outputs were overwritten"
            kept = (out["info"].means2d, out["info"].means2d.clone())
            losses.append(loss)
            images.append((out["rgb"], out["rgb"].clone()))
        torch.cuda.synchronize()
        for a, b in images:
            assert torch.equal(a, b), "an image the caller kept was overwritten by a later step"
        finals[express] = [float(x) for x in losses]
        assert len(set(finals[express])) > 10  # (every step's own value)
        assert (tr.express_steps == 12) if express else (tr.express_steps == 0)
        assert tr.bin_frames.get("fill", 0) == 12 and tr.live_overflows == 0
    for it, (x, y) in enumerate(zip(finals[True], finals[False])):
        assert math.isfinite(x) and abs(x - y) <= (1e-5 if it < 4 else 1e-3) * abs(y), (it, x, y)


def test_express_overflow_is_redone_by_the_per_op_route(dev):
    """Buckets shrunk to a quarter: fsgs_step_run returns FSGS_STEP_OVERFLOW before any launch with side effects, the
    trainer redoes the frame through the exact two-pass route — one Adam step, the images of the untouched route."""
    from fusionsense_amd.trainer import SplatTrainer
    params, cams, tgts = _scene(dev, n=60_000, seed=9)
    tr = SplatTrainer(params, dev, sh_degree=3, seed=3)
    for it in range(6):
        tr.train_step(cams[it % 3], tgts[it % 3])
    snap = _snapshot(tr)
    tr.express = False
    _, ref = tr.train_step(cams[2], tgts[2])
    ref_rgb, ref_n = ref["rgb"].clone(), int(ref["info"].n_live)
    _restore(tr, snap)
    tr.express = True
    for h in tr._bin_hist.values():
        h["base"] = (h["base"].float() * 0.25).to(torch.int32)
    steps0, over0, adam0 = tr.express_steps, tr.live_overflows, tr.adam_steps
    _, out = tr.train_step(cams[2], tgts[2])
    assert tr.live_overflows == over0 + 1 and tr.express_steps == steps0 and tr.adam_steps == adam0 + 1
    assert torch.equal(out["rgb"], ref_rgb) and int(out["info"].n_live) == ref_n
    assert out["info"].bin_route == "two_pass"
    # the redone frame left exact room: the view's next visit is a one-call step again
    tr.train_step(cams[2], tgts[2])
    assert tr.express_steps == steps0 + 1


def test_express_rebuilds_after_the_model_changed_and_times_launches(dev):
    """Pruning re-creates parameters and moments: the argument blocks (raw pointers) are rebuilt, never reused; and
    ops.TIMER's spans (bench.py's per-kernel timing) are recorded by the library around the launches it asks for."""
    from fusionsense_amd import ops
    from fusionsense_amd.trainer import PARAM_ORDER, SplatTrainer
    params, cams, tgts = _scene(dev, n=40_000, seed=13)
    tr = SplatTrainer(params, dev, sh_degree=3, seed=3)
    for it in range(6):
        tr.train_step(cams[it % 3], tgts[it % 3])
    ex0 = tr._express
    assert ex0 is not None
    # drop every tenth Gaussian by hand, the way densify / prune surgery does: new parameter and moment tensors
    keep = torch.ones(tr.num_gaussians(), dtype=torch.bool, device=dev)
    keep[::10] = False
    for name in PARAM_ORDER:
        old = tr._params[name]
        st = tr.optimizers[name].state.pop(old)
        new = torch.nn.Parameter(old.data[keep].contiguous())
        tr._params[name] = new
        tr.optimizers[name].param_groups[0]["params"] = [new]
        tr.optimizers[name].state[new] = {"step": st["step"], "exp_avg": st["exp_avg"][keep].contiguous(),
                                          "exp_avg_sq": st["exp_avg_sq"][keep].contiguous()}
    tr.slab.rebuild(tr._params)
    for it in range(6):
        loss, _ = tr.train_step(cams[it % 3], tgts[it % 3])
    assert math.isfinite(float(loss)) and tr._express is not ex0 and tr._express.N == int(keep.sum())
    ops.TIMER.reset(enabled=True, only=("raster_bwd", "gaussian_bwd"), prealloc=8)
    try:
        before = tr.express_steps
        for it in range(3):
            tr.train_step(cams[it % 3], tgts[it % 3])
        assert tr.express_steps == before + 3
        summ = ops.TIMER.summary()
    finally:
        ops.TIMER.reset(enabled=False)
    assert summ["raster_bwd_quad_d4e3"]["calls"] == 3 and summ["gaussian_bwd"]["calls"] == 3
    assert 0.0 < summ["raster_bwd_quad_d4e3"]["avg_ms"] < 50.0


def test_handoff_timeout_word_is_looked_at(dev):
    """a bounded wait of the forward's hand-off protocol that runs into its bound sets TailHeader.error; the
    trainer's periodic demand probe now carries that word and raises — the hand-off is switched off — instead of
    training on with possibly wrong frames."""
    from fusionsense_amd import fused
    w = fused.FWD_WALK
    saved = (w.handoff_records, w.failed)
    w.state.pop(str(dev), None)
    st = w._state(dev)
    st["scratch"][8:12].view(torch.int32).fill_(1)  # what raster_quad.hip's timed-out wait leaves
    try:
        with pytest.raises(RuntimeError, match="hand-off"):
            for _ in range(3 * w.PROBE_EVERY + 2):
                w.scratch(dev)
                torch.cuda.synchronize()
        assert w.handoff_records == 0 and w.failed and w.choice(1000)[1] == 0
    finally:
        w.handoff_records, w.failed = saved
        w.state.pop(str(dev), None)
