"""A minimal training loop around the hot path, for measurement and for multi-GPU data
parallelism.  It mirrors the iteration order of nerfstudio's Trainer
(SURVEY.md A.3): zero_grad -> get_outputs -> losses -> backward -> [grad all-reduce] ->
Adam step -> after_train statistics -> refinement_after every ``refine_every`` steps, with the
8 Adam groups of /root/reference/dn_splatter/dn_config.py:36-75.

Multi-GPU: one process per GPU, every rank holds a full replica of the Gaussians and renders
a different training view (views r, r+W, r+2W, ...).  Gradients of all parameter groups live in
ONE contiguous fp32 slab, so the exchange step is a single RCCL all-reduce over xGMI
(236 B/Gaussian) plus one 12 B/Gaussian all-reduce of the densification statistics; densify/
prune then runs redundantly with a shared RNG seed so N stays identical on every rank.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
from torch import Tensor

from . import comm
from .scenes import Camera

PARAM_ORDER = ["means", "scales", "quats", "features_dc", "features_rest", "opacities"]
# The gradient slab keeps the two SH feature groups (48 of the 59 floats per Gaussian) at its end, so that the
# geometry half (what projection and binning of the NEXT frame need) and the feature half are one contiguous
# range each: see SplatTrainer._reduce_and_step.
GEOMETRY_GROUPS = ["means", "scales", "quats", "opacities"]
FEATURE_GROUPS = ["features_dc", "features_rest"]
SLAB_ORDER = GEOMETRY_GROUPS + FEATURE_GROUPS
HALF_GROUPS = ["scales", "quats", "features_dc", "features_rest", "opacities"]  # config #5: stored half for rendering


@dataclass
class OptimConfig:
    """dn_config.py:36-75 (the 'normals' and 'camera_opt' groups are inert: SURVEY.md A.2/A.3)."""
    lr: Dict[str, float] = field(default_factory=lambda: {
        "means": 1.6e-4, "features_dc": 0.0025, "features_rest": 0.0025 / 20, "opacities": 0.05,
        "scales": 0.005, "quats": 0.001})
    eps: float = 1e-15
    means_lr_final: float = 1.6e-6
    means_lr_max_steps: int = 30000


def _adam_range_torch(p: Tensor, g: Tensor, m: Tensor, v: Tensor, lr: float, step: int, b1: float, b2: float,
                      eps: float) -> None:
    """torch.optim.Adam's single-tensor update (amsgrad off, no weight decay) on flat ranges — the host-side twin of
    fsgs_adam_step for the CPU (gloo) tests of the sharded step."""
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


class GradSlab:
    """All parameter gradients as views into one flat fp32 buffer (one collective per step).

    ``align`` / ``world`` (sharded geometry step, SplatTrainer._reduce_and_step): every geometry group then starts at a
    multiple of ``align`` floats and the geometry half's length is a multiple of ``align * world``, so that the half
    splits into ``world`` equal shards whose intersections with the groups are 16-byte aligned ranges; ``geo_ranges``
    lists the groups' (name, start, numel) inside the half.  Padding floats stay zero."""

    def __init__(self, params: Dict[str, torch.nn.Parameter], align: int = 1, world: int = 1, on_rebuild=None,
                 after_rebuild=None):
        self.align, self.world = int(align), int(world)
        self.on_rebuild, self.after_rebuild = on_rebuild, after_rebuild
        self.rebuild(params)

    def rebuild(self, params: Dict[str, torch.nn.Parameter]) -> None:
        if self.on_rebuild is not None:
            self.on_rebuild(params)  # (may change align / world: the trainer decides the layout from N)
        up = lambda x, a: (x + a - 1) // a * a  # noqa: E731
        offs, off = {}, 0
        for name in GEOMETRY_GROUPS:
            off = up(off, self.align)
            offs[name] = off
            off += params[name].numel()
        self.split = up(off, self.align * self.world)  # first element of the feature half
        self.geo_ranges = [(name, offs[name], params[name].numel()) for name in GEOMETRY_GROUPS]
        off = self.split
        for name in FEATURE_GROUPS:
            offs[name] = off
            off += params[name].numel()
        dev = next(iter(params.values())).device
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.views = {}
        for name in SLAB_ORDER:
            p = params[name]
            v = self.flat[offs[name]:offs[name] + p.numel()].view_as(p)
            self.views[name] = v
            p.grad = v
        if self.after_rebuild is not None:
            self.after_rebuild()  # (the trainer re-creates its flat parameter / moment slabs for the new layout)

    def zero_(self) -> None:
        self.flat.zero_()

    # (every collective of the step goes through fusionsense_amd/comm.py: one code path for RCCL and gloo)
    @staticmethod
    def _world(group=None) -> int:
        return comm.world(group)

    @staticmethod
    def _exchange(group=None) -> bool:
        """Whether a collective has to be issued at all."""
        return comm.exchange(group)

    def _reduce_mean(self, t: Tensor, group, async_op: bool):
        """Mean over the ranks, in place.  Returns None (done, in stream order) or a callable that completes it."""
        return comm.all_reduce_mean_(t, group, async_op)

    def all_reduce_mean_(self, group=None) -> None:
        self._reduce_mean(self.flat, group, False)

    def all_reduce_geometry_mean_(self, group=None) -> None:
        self._reduce_mean(self.flat[:self.split], group, False)

    def all_reduce_features_async(self, group=None):
        """Starts the exchange of the feature half; the returned callable (or None) completes it."""
        return self._reduce_mean(self.flat[self.split:], group, True)


class SplatTrainer:
    def __init__(self, params: Dict[str, Tensor], device: torch.device, sh_degree: int = 3,
                 optim: Optional[OptimConfig] = None, fused_adam: bool = True, seed: int = 0,
                 strategy=None, fused: bool = True, sh_degree_interval: Optional[int] = None,
                 direct: bool = True, loss_cfg=None, half_attributes: bool = False, torch_optimizers: bool = False,
                 fuse_torch_optimizers: bool = False):
        self.device = device
        # fused=True: get_outputs as one autograd node (fusionsense_amd/fused.py); False: the
        # reference's op-by-op caller through the drop-in rasterization()/rasterize_gaussians() surface
        self.fused = fused
        # direct=True (fused only): the step's fixed two-node graph (get_outputs, loss) is run forward and
        # backward by hand instead of through torch.autograd (same kernels, same gradients in the slab)
        self.direct = direct
        self.sh_degree = sh_degree
        # The loss.  A step's supervision is either a prepared ``losses.FrameBatch`` (image, sensor depth, mono
        # normals, mask): then the loss is the reference's get_loss_dict (dn_model.py:673-925) with the switches of
        # ``loss_cfg`` (losses.LossConfig; default = FusionSense's configs/config.py) — or a plain dict of target
        # images rgb / depth / normal: the BENCHMARK loss of BASELINE config #2 (L1 + SSIM on rgb,
        # plain L1 on depth and normals over every pixel), which is a measurement harness and not get_loss_dict.
        if loss_cfg is None:
            from .losses import LossConfig
            loss_cfg = LossConfig()
        self.loss_cfg = loss_cfg
        # sh_degree_to_use = min(step // sh_degree_interval, sh_degree) (dn_model.py:562-565; nerfstudio default
        # interval 1000); None = always the full degree (BASELINE config #2 is quoted at degree 3)
        self.sh_degree_interval = sh_degree_interval
        self.optim_cfg = optim or OptimConfig()
        # (a copy: training must not write into the caller's tensors)
        self._params: Dict[str, torch.nn.Parameter] = {
            k: torch.nn.Parameter(params[k].detach().to(device=device, dtype=torch.float32, copy=True).contiguous())
            for k in PARAM_ORDER}
        self.fused_adam = fused_adam and device.type == "cuda"
        # torch_optimizers: step the eight torch.optim.Adam objects themselves (what nerfstudio's Optimizers does)
        # instead of the one-launch libfsgs Adam — the configuration integration.patch() runs under
        self.torch_optimizers = bool(torch_optimizers)
        self.optimizers: Dict[str, torch.optim.Adam] = {}
        for name in PARAM_ORDER:
            self.optimizers[name] = torch.optim.Adam(
                [self._params[name]], lr=self.optim_cfg.lr[name], eps=self.optim_cfg.eps,
                fused=self.fused_adam)
        # ... and with fuse_torch_optimizers their step() calls are gathered into one libfsgs launch, the optimizer
        # objects and their state untouched (integration.AdamFuser: what patch_all offers under nerfstudio's Optimizers)
        self.adam_fuser = None
        if self.torch_optimizers and fuse_torch_optimizers:
            from .integration import fuse_optimizers
            self.adam_fuser = fuse_optimizers(self.optimizers, PARAM_ORDER)
        # Sharded geometry step (data-parallel runs of large scenes): instead of all-reducing the geometry
        # half of the slab and running Adam on all of it on every rank, the half is REDUCE-SCATTERED (each rank receives
        # the mean gradient of its 1/W shard), Adam runs on the owned shard only, and the updated PARAMETERS are
        # all-gathered — the same bytes on the links as the all-reduce, 1/W of the geometry's Adam traffic per rank
        # (28 B per parameter float), and replicas that are bit-identical by construction.  The four geometry parameters
        # and their Adam moments then live in flat, padded slabs with the gradient slab's layout (``_geo``).
        # FSGS_SHARDED_GEOMETRY: "auto" (default) = with several ranks and at least 2^20 Gaussians, "1" = whenever a
        # group exists, "0" = never.
        self.sharded_geometry_mode = os.environ.get("FSGS_SHARDED_GEOMETRY", "auto")
        self.sharded_min_n = 1 << 20
        self._geo = None            # dict(P, M, V flat slabs, shard bounds) while the sharded layout is active
        self._moments_local = False  # moments outside the owned shard are stale until sync_optimizer_state()
        self.half_attributes = bool(half_attributes)
        self.slab = None
        self.slab = GradSlab(self._params, on_rebuild=self._choose_layout, after_rebuild=self._flatten_geometry)
        if self._want_sharded(self.num_gaussians()):
            self.slab.rebuild(self._params)  # (now that the slab exists: the padded layout + the flat slabs)
        # Data-parallel runs: the SH features' share of the all-reduce (81 % of the slab) and their Adam launch are
        # deferred until the next frame needs the colours, i.e. they overlap that frame's projection, binning and
        # sort, which read geometry only (exact: every parameter still sees its fully reduced gradient before its
        # next use).  FSGS_DEFER_FEATURES=0 switches it off; FSGS_FORCE_SPLIT_STEP=1 takes the same code path on a
        # single rank (for the tests).
        self.defer_features = os.environ.get("FSGS_DEFER_FEATURES", "1") != "0"
        self.force_split_step = os.environ.get("FSGS_FORCE_SPLIT_STEP", "0") == "1"
        # One GPU: the SH features' Adam launch (48 of the 59 floats per Gaussian, bandwidth-bound) is deferred too and
        # RIDES in the next frame's projecting count pass (vector-ALU / latency-bound; reads geometry only) — one
        # launch with two independent block ranges instead of two launches back to back.  Same arithmetic, same
        # order as seen by every reader (flush() lands a pending update for anything else that reads the features).
        self.ride_adam = os.environ.get("FSGS_RIDE_ADAM", "1") != "0"
        # (only while the count pass runs as one chunk per 1024-thread workgroup, 256 k .. 512 k Gaussians: the larger
        # instantiation needs 78 VGPRs, one workgroup per CU, and the riding Adam blocks then crawl: 4.0 instead of
        # 0.8 + 1.4 ms at 6 M Gaussians; smaller scenes have nothing to hide.  For large scenes the same update on a
        # second HIP stream beside the next frame's binning was measured too: 6.44 against 6.41 ms at 6 M, 9.41 against
        # 9.44 ms at 10 M Gaussians — both sides are bandwidth-sensitive there and slow each other down; not kept.)
        self.ride_adam_n = (256 * 1024, 512 * 1024)
        # ... and what travels for the features is not their gradient (48 floats per Gaussian) but its factors: the
        # masked colour gradient + the camera centre of every rank's view (4 floats per Gaussian and rank,
        # all-gathered); fsgs_sh_coeff_grad rebuilds the mean gradient on every rank, in rank order, so the
        # replicas stay bit-identical.  Fused CUDA path only; FSGS_FACTORED_FEATURES=0 switches it off.
        self.factored_features = os.environ.get("FSGS_FACTORED_FEATURES", "1") != "0"
        # BASELINE config #5: the render kernels read IEEE-half MIRRORS of every attribute but the means (SH features,
        # log-scales, quaternions, opacity logits: 124 instead of 236 parameter bytes per Gaussian per frame); the
        # fp32 masters, their gradients and Adam are untouched, and the fused Adam launch rewrites the mirrors.
        self.half_attributes = bool(half_attributes)
        self._half: Optional[Dict[str, Tensor]] = None
        self._half_dirty = True
        if self.half_attributes:
            assert fused and device.type == "cuda", "half attribute storage runs on the fused HIP path"
            self.factored_features = False  # (the factored exchange rebuilds from fp32 features)
        # No host wait inside the tape-free step: the frame's list buffers are sized from the largest live-pair count
        # this (N, tiles) shape has shown so far (+ 25 %), the kernels take the true total from the device, and the
        # host looks at it only after the forward's launches are enqueued; a frame that exceeds the estimate is
        # redone with exact sizes (ops.LiveListOverflow).  FSGS_NO_WAIT=0 restores the wait in front of the bucket fill.
        self.no_wait = os.environ.get("FSGS_NO_WAIT", "1") != "0"
        from .capacity import LiveCapacity
        self._live_caps = LiveCapacity()  # (bucketed N, windowed maximum, LRU-bounded: capacity.py)
        self.live_overflows = 0
        # Occlusion cuts: in a dense scene the forward walk stops long before the end of a tile's
        # depth-ordered list (measured at 10 M Gaussians: 2 % of the sorted pairs are ever examined), so the pairs
        # behind the depth at which the previous frame OF THE SAME VIEW saturated (+ a margin) are not binned, sorted
        # or staged at all.  Exactness is kept by a check, not by the estimate: a cut tile that does not saturate
        # inside its prefix spoils the frame (ops.OcclusionCutInvalid) and the frame is redone without cuts.
        # FSGS_OCCLUSION_CUT = auto (N >= occlusion_cut_min_n) | 1 | 0;  margins: FSGS_ZCUT_MARGIN="span,rel".
        self.occlusion_cut_mode = os.environ.get("FSGS_OCCLUSION_CUT", "auto")
        self.occlusion_cut_min_n = 1 << 20
        # The next cut of a tile = the depth of its deepest composited entry + max(span x (that depth - the depth of the
        # tile's first entry), rel x that depth).  Measured on config #4 (8 views, benchmark loss on random targets,
        # i.e. large steps): (0.5, 0.01) keeps 1.6 M of 39 M pairs and 4 of 58 cut frames are redone, (1.0, 0.03)
        # keeps 2.7 M and none is redone, (2.0, 0.06) keeps 8.3 M.  A view whose frame had to be redone doubles its own
        # margins (up to 8x).
        # Third number (tail): a tile that saturated only within the last quarter of its list gets no cut — it gains little
        # and is the tile found open one visit later (config #4: 17 → 3 second looks per 58 cut frames, same pairs kept).
        m = (os.environ.get("FSGS_ZCUT_MARGIN", "1.0,0.03,0.25") + ",0.25").split(",")
        self.zcut_margins = (float(m[0]), float(m[1]), float(m[2]))
        self._zcuts: Dict = {}
        self._zcut_widen: Dict = {}
        self._bin_hist: Dict = {}   # view -> {"base": bucket bases [T + 1], "n_live": pairs} of its previous frame
        self._bin_hist_n = -1
        # A model whose size moved by more than this fraction since a view's bucket room was noted starts its views over
        # (their next visit takes the exact two-pass route and leaves fresh room).  Round 6 measured the alternative under
        # densification (`bench.py --config 2 --steps 300 --densify`, ops.BIN_FILL_GROWTH): dropping the room at every
        # refinement (0.02) makes every frame of a 100-view epoch a two-pass frame (0.62 ms per step); keeping it with
        # buckets grown by 2.0 instead of 1.25 keeps the count-free route without a single redone frame (0.58).
        self.bin_hist_keep_rel = float(os.environ.get("FSGS_BIN_HIST_KEEP_REL", "0.10"))
        self.bin_frames: Dict = {}  # frames by binning route ("fill": count-free, "two_pass")
        self.zcut_max_views = 1024  # least recently rendered views beyond this are forgotten (T floats each)
        # Adam in the backward: on one rank, when nothing else needs the gradients, the per-Gaussian
        # backward launch applies the step's Adam update itself (fsgs_gauss_sh_bwd_adam) — the 236 B of gradients per
        # Gaussian are neither written nor read back, and the Adam launch disappears.  The gradient slab is then NOT
        # filled by that step.  Measured: +0.9 % at 100 k Gaussians, +2.7 % at config #3, +7 % at 700 k, +13-15 % at
        # configs #4 / #5, +0.6 % at config #2 (against the feature update riding in the next count pass, which remains
        # the route of FSGS_ADAM_IN_BACKWARD=0 for ride_adam_n sizes).
        # FSGS_ADAM_IN_BACKWARD = auto (whenever the step qualifies) | 1 (the same) | 0.
        self.adam_in_backward_mode = os.environ.get("FSGS_ADAM_IN_BACKWARD", "auto")
        self.adam_in_backward_min_n = 0
        self.adam_in_backward_steps = 0
        self.cut_frames = 0
        self.cut_redone = 0
        self._factors = None
        self._pending = None
        # False after a step whose Adam update ran inside the backward launch (no gradients written: gradients() raises,
        # every parameter's .grad is None); True after any step that left gradients in the slab
        self.last_step_grads_valid = True
        # express.py: qualifying steps are enqueued by ONE library call (fsgs_step_run).  FSGS_EXPRESS=0 / .express = False
        # keeps the launch-by-launch route everywhere (A/B, tests of that route)
        self.express = os.environ.get("FSGS_EXPRESS", "1") != "0"
        self._express = None
        self.express_steps = 0
        self.comm_events = None  # a list: (step, start, end) HIP-event pairs around every collective / wait (see _comm)
        self.step = 0
        self.strategy = strategy  # fusionsense_amd.splatfacto.DensifyStrategy or None
        if strategy is not None and hasattr(strategy, "stage"):
            strategy.stage(device)  # (the callbacks' scene metadata: on the device once, not inside every callback)
        self.rng = torch.Generator(device=device)
        self.rng.manual_seed(seed)
        self.last_info = None

    # -- exposed communication time --------------------------------------------------------------------------------
    def _comm(self):
        """Context manager around a collective (or the wait for one) on the step's stream: with ``comm_events`` set to a
        list, an event pair is recorded around it — the time the compute stream spends inside / blocked by the collective,
        i.e. communication that was NOT hidden behind compute (bench.py: comm_ms_exposed)."""
        import contextlib
        ev = self.comm_events
        if ev is None or self.device.type != "cuda":
            return contextlib.nullcontext()
        trainer = self

        @contextlib.contextmanager
        def span():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            try:
                yield
            finally:
                b.record()
                ev.append((trainer.step, a, b))
        return span()

    # -- sharded geometry step -------------------------------------------------------------------------------------
    def _want_sharded(self, n: int) -> bool:
        mode = getattr(self, "sharded_geometry_mode", "0")
        if mode == "0" or self.half_attributes or not GradSlab._exchange():
            return False
        return mode == "1" or (GradSlab._world() > 1 and n >= self.sharded_min_n)

    def _choose_layout(self, params) -> None:
        """GradSlab.on_rebuild: the slab's layout follows the step the trainer will run for this N."""
        if self.slab is None:
            return  # (first build: GradSlab.__init__ is still running; _init_layout below sets it and rebuilds)
        sh = self._want_sharded(params["means"].shape[0])
        self.slab.align, self.slab.world = (64, GradSlab._world()) if sh else (1, 1)

    def _flatten_geometry(self) -> None:
        """(Re-)creates the flat parameter / moment slabs of the geometry half after the parameters were (re)built:
        every geometry parameter becomes a view into ``P`` and its Adam moments views into ``M`` / ``V``, at the
        gradient slab's offsets.  No-op (and back to ordinary tensors) when the sharded step is not in use."""
        sl = self.slab
        if sl is None or sl.align == 1:
            self._geo = None
            self._moments_local = False
            return
        dev = sl.flat.device
        L, W, r = sl.split, sl.world, (comm.rank() if GradSlab._exchange() else 0)
        P, M, V = (torch.zeros(L, dtype=torch.float32, device=dev) for _ in range(3))
        for name, off, n in sl.geo_ranges:
            old = self._params[name]
            opt = self.optimizers[name]
            st = opt.state.pop(old, None)
            P[off:off + n].copy_(old.data.reshape(-1))
            new = torch.nn.Parameter(P[off:off + n].view_as(old))
            new.grad = sl.views[name]
            st = st if st else {"step": torch.tensor(0.0)}
            for key, flat in (("exp_avg", M), ("exp_avg_sq", V)):
                if key in st:
                    flat[off:off + n].copy_(st[key].reshape(-1))
                st[key] = flat[off:off + n].view_as(old)
            opt.state[new] = st
            opt.param_groups[0]["params"] = [new]
            self._params[name] = new
        shard = L // W
        self._geo = dict(P=P, M=M, V=V, shard=shard, lo=r * shard, hi=(r + 1) * shard,
                         grad=torch.zeros(shard, dtype=torch.float32, device=dev))
        self._moments_local = False

    def sync_optimizer_state(self) -> None:
        """Sharded geometry step: every rank has updated the Adam moments of its own shard only; before anything reads
        or moves whole moment tensors (densification surgery, opacity reset, checkpoints) the shards are all-gathered.
        A collective: every rank calls it at the same point (the callers are deterministic in the step number)."""
        g = self._geo
        if g is None or not self._moments_local:
            return
        with self._comm():
            for flat in (g["M"], g["V"]):
                self._all_gather_shards(flat, g)
        self._moments_local = False

    def _all_gather_shards(self, flat: Tensor, g) -> None:
        if not GradSlab._exchange():
            return
        comm.all_gather_shards_(flat, g["lo"], g["hi"])  # (in place: the input is the output's own slice)

    def _sharded_geometry_step(self) -> int:
        """reduce-scatter (mean) of the geometry half -> Adam on the owned shard -> all-gather of the parameters."""
        g, sl = self._geo, self.slab
        W = sl.world
        geo = sl.flat[:sl.split]
        if GradSlab._exchange():
            with self._comm():
                comm.reduce_scatter_mean(g["grad"], geo)
        else:
            g["grad"].copy_(geo[g["lo"]:g["hi"]])
        self.adam_steps = getattr(self, "adam_steps", 0) + 1
        step_no = self.adam_steps
        ps, gs, ms, vs, lrs = [], [], [], [], []
        for name, off, n in sl.geo_ranges:
            st = self.optimizers[name].state[self._params[name]]
            st["step"] += 1
            a, b = max(off, g["lo"]), min(off + n, g["hi"])
            if a >= b:
                continue
            ps.append(g["P"][a:b]); gs.append(g["grad"][a - g["lo"]:b - g["lo"]])
            ms.append(g["M"][a:b]); vs.append(g["V"][a:b])
            lrs.append(self.optimizers[name].param_groups[0]["lr"])
        if ps:
            if self.device.type == "cuda":
                from .ops import adam_step_
                adam_step_(ps, gs, ms, vs, lrs, step_no, 0.9, 0.999, self.optim_cfg.eps)
            else:
                for p_, g_, m_, v_, lr in zip(ps, gs, ms, vs, lrs):
                    _adam_range_torch(p_, g_, m_, v_, lr, step_no, 0.9, 0.999, self.optim_cfg.eps)
        with self._comm():
            self._all_gather_shards(g["P"], g)
        self._moments_local = W > 1
        return step_no

    # -- schedule: ExponentialDecayScheduler on the means group (dn_config.py:38-41) --------
    def _means_lr(self, step: int) -> float:
        c = self.optim_cfg
        t = min(max(step / c.means_lr_max_steps, 0.0), 1.0)
        return math.exp(math.log(c.lr["means"]) * (1 - t) + math.log(c.means_lr_final) * t)

    @property
    def params(self) -> Dict[str, torch.nn.Parameter]:
        """The parameter dictionary as every OUTSIDE reader must see it: a deferred feature update (data-parallel
        runs; on one GPU the features' Adam step that rides in the next frame's count pass, 256 k .. 512 k Gaussians)
        is landed first, so exports, EMA copies, user scripts and the densification callbacks never read features
        that are one Adam step behind.  The step itself uses ``_params`` (no flush)."""
        self.flush()
        return self._params

    def num_gaussians(self) -> int:
        return self._params["means"].shape[0]

    def half_mirrors(self) -> Optional[Dict[str, Tensor]]:
        """The float16 mirrors of HALF_GROUPS (None unless half_attributes), re-derived from the masters whenever
        something other than the fused Adam launch has written the parameters (densify / prune, resets, loads)."""
        if not self.half_attributes:
            return None
        stale = self._half is None or self._half_dirty or any(
            self._half[k].shape != self._params[k].shape for k in HALF_GROUPS)
        if stale:
            self._half = {k: self._params[k].data.to(torch.float16).contiguous() for k in HALF_GROUPS}
            self._half_dirty = False
        return self._half

    def mark_params_written(self) -> None:
        """Tell the trainer that parameter data changed outside the optimizer step (mirrors must be re-derived)."""
        self._half_dirty = True

    def _sh_degree_now(self, sh_degree_to_use: Optional[int] = None) -> int:
        if sh_degree_to_use is not None:
            return sh_degree_to_use
        if self.sh_degree_interval:
            return min(self.step // self.sh_degree_interval, self.sh_degree)
        return self.sh_degree

    def _frame_state(self, camera: Camera, grad: bool):
        """(after_train statistics target, touch-anchor mask, binary-opacity threshold or None) for this step:
        what the model state supplies to get_outputs in the reference (dn_model.py:492-503, 535-541)."""
        stats, add_mask, bthr = None, None, None
        if self.strategy is not None:
            add_mask = getattr(self.strategy, "add_mask", None)
            cfg = getattr(self.strategy, "cfg", None)
            if cfg is not None:
                from .splatfacto import binary_opacity_active
                if binary_opacity_active(cfg, self.step):
                    bthr = cfg.binary_opacities_threshold
            if grad and self.fused and hasattr(self.strategy, "stats_target"):
                stats = self.strategy.stats_target(self, self.num_gaussians(), self.device, camera)
        return stats, add_mask, bthr

    def forward(self, camera: Camera, sh_degree_to_use: Optional[int] = None, crop_box=None):
        """get_outputs.  In training mode (grad enabled) the strategy supplies what the model state supplies
        in the reference: the binary-opacity schedule (dn_model.py:492-503) and the touch-anchor mask
        (:535-541); ``crop_box`` is honoured only without grad, like ``not self.training`` there (:505-532)."""
        self.flush()
        deg = self._sh_degree_now(sh_degree_to_use)
        grad = torch.is_grad_enabled()
        stats, add_mask, bthr = self._frame_state(camera, grad)
        if self.fused:
            from .fused import render_fusionsense_fused
            return render_fusionsense_fused(self._params, camera, sh_degree=deg, device=self.device,
                                            grad_out=self.slab.views if grad else None, stats_out=stats,
                                            add_mask=add_mask, crop_box=crop_box, training=grad,
                                            binary_threshold=bthr, half=self.half_mirrors())
        from .fusion import render_fusionsense
        return render_fusionsense(self._params, camera, sh_degree=deg, device=self.device, add_mask=add_mask,
                                  crop_box=crop_box, training=grad, binary_threshold=bthr)

    def _touch_rows(self):
        """(row indices of the touch anchors, their contact normals) for the touch-normal MSE (dn_model.py:893-902)."""
        st = self.strategy
        if st is None or getattr(st, "add_mask", None) is None or getattr(st, "touch_normals", None) is None:
            return None, None
        return st.touch_rows(), st.touch_normals

    def loss(self, out, target, camera: Optional[Camera] = None) -> Tensor:
        """FrameBatch -> get_loss_dict (dn_model.py:673-925).  Dict of target images -> the config-#2 benchmark loss
        (SURVEY.md §8d): 0.8*L1 + 0.2*(1-SSIM) on rgb, L1 on depth, L1 on normals."""
        from .losses import FrameBatch, rgb_loss
        if isinstance(target, FrameBatch):
            from .ops import fusion_loss
            ti, tn = self._touch_rows()
            # The fused get_outputs node WRITES its gradients into the slab (it does not accumulate), so on that route
            # the min-scale term's direct gradient to the log-scales is added by train_step after the backward
            # instead of travelling through autograd's accumulation, whose order against the node's write is open.
            scales = self._params["scales"].detach() if self.fused else self._params["scales"]
            intr = (camera.fx, camera.fy, camera.cx, camera.cy) if camera is not None else None
            return fusion_loss(out, target, self.loss_cfg, scales, ti, tn, intr)
        if self.fused and "depth" in target and out["rgb"].is_cuda:
            from .ops import train_loss
            has_n = "normal" in target
            return train_loss(out["rgb"], target["rgb"], out["depth"], target["depth"],
                              out["normal"] if has_n else None, target["normal"] if has_n else None, 0.2, 0.2, 0.1)
        l = rgb_loss(out["rgb"], target["rgb"])
        if "depth" in target:
            l = l + 0.2 * torch.abs(out["depth"] - target["depth"]).mean()
        if "normal" in target:
            l = l + 0.1 * torch.abs(out["normal"] - target["normal"]).mean()
        return l

    def _fused_adam_step(self, names=None, step_no: Optional[int] = None, as_groups: bool = False):
        """All six groups in one launch (row N1) — or the given subset, as Adam step ``step_no``.  The torch
        optimizers stay the owners of the state (exp_avg / exp_avg_sq per parameter), so densify/prune surgery
        and checkpoints see the usual layout; only their step() is replaced.  Returns the step number used."""
        from .ops import adam_groups, adam_step_
        ps, gs, ms, vs, lrs = [], [], [], [], []
        for name in (names or PARAM_ORDER):
            p = self._params[name]
            opt = self.optimizers[name]
            st = opt.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
            if not as_groups:  # (a carried step is counted when its launch has been enqueued: _adam_rider)
                st["step"] += 1
            ps.append(p.data); gs.append(self.slab.views[name]); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"])
            lrs.append(opt.param_groups[0]["lr"])
        if step_no is None:
            self.adam_steps = getattr(self, "adam_steps", 0) + 1
            step_no = self.adam_steps
        hm = self.half_mirrors()
        halves = [hm.get(name) for name in (names or PARAM_ORDER)] if hm is not None else None
        if as_groups:  # (not launched here: the argument block for a launch that carries it, see _adam_rider)
            return adam_groups(ps, gs, ms, vs, lrs, step_no, 0.9, 0.999, self.optim_cfg.eps, half_mirrors=halves)
        adam_step_(ps, gs, ms, vs, lrs, step_no, 0.9, 0.999, self.optim_cfg.eps, half_mirrors=halves)
        return step_no

    def _optimizer_step(self, names, step_no: Optional[int] = None) -> Optional[int]:
        if self.fused and self.device.type == "cuda" and not self.torch_optimizers:
            return self._fused_adam_step(names, step_no)
        for name in names:
            self.optimizers[name].step()
        if self.adam_fuser is not None:
            self.adam_fuser.flush()  # (a subset of the groups was stepped: launch what is pending)
        if self.half_attributes:
            self.mark_params_written()  # torch's step does not rewrite the half mirrors: re-derive before the next frame
        return None

    def _ride_mode(self) -> bool:
        """One rank, the library's own Adam, the tape-free step: the feature update can ride in the count pass."""
        # (... unless the whole Adam step can go into the backward launch, §9.9: measured at config #2, 0.584 against
        # 0.588 ms per step — the count pass no longer carries 400 MB of feature moments, the per-Gaussian backward
        # carries all six groups; riding remains what FSGS_ADAM_IN_BACKWARD=0 gets)
        return (self.ride_adam and self._one_rank_fused() and self.adam_in_backward_mode == "0" and
                self.ride_adam_n[0] <= self.num_gaussians() <= self.ride_adam_n[1])

    def _one_rank_fused(self) -> bool:
        return (GradSlab._world() == 1 and not self.force_split_step and self.fused and self.direct and self.fused_adam
                and not self.torch_optimizers and self.device.type == "cuda")

    def _split_step(self, optimizer_step: bool) -> bool:
        return optimizer_step and self.defer_features and (GradSlab._world() > 1 or self.force_split_step
                                                           or self._ride_mode())

    def _adam_rider(self):
        """The pending feature update as something the next frame's count launch can carry (fused.FrameInfo.adam_rider),
        or None when there is none or it needs more than an Adam launch (an exchange still in flight, factors)."""
        if self._pending is None or not self._ride_mode():
            return None
        finish, step_no = self._pending
        if finish is not None:
            return None
        trainer = self

        class _Rider:
            groups = trainer._fused_adam_step(FEATURE_GROUPS, step_no, as_groups=True)

            @staticmethod
            def consumed():
                trainer._pending = None
                for name in FEATURE_GROUPS:
                    trainer.optimizers[name].state[trainer.params[name]]["step"] += 1
        return _Rider

    def _factor_buffers(self, optimizer_step: bool):
        """(own [N + 1,4], gathered [R, N + 1,4]) when this step exchanges the SH gradients as factors, else None."""
        if not (self._split_step(optimizer_step) and self.factored_features and self.fused
                and self.device.type == "cuda") or self._ride_mode():
            return None
        R, N = GradSlab._world(), self.num_gaussians()
        f = self._factors
        if f is None or f[1].shape[0] != R or f[1].shape[1] != N + 1:
            gathered = torch.empty(R, N + 1, 4, dtype=torch.float32, device=self.device)
            # a single rank writes straight into the gathered buffer
            own = gathered[0] if R == 1 else torch.empty(N + 1, 4, dtype=torch.float32, device=self.device)
            self._factors = f = (own, gathered)
        return f

    def _features_from_factors(self, sh_degree: int, step_no: int) -> None:
        """The mean SH coefficient gradient of the gathered views, rebuilt AND applied (Adam step ``step_no`` of the
        two feature groups) in one launch: the gradient itself is never written (fsgs_sh_coeff_grad_adam)."""
        from ._lib import load, ptr, stream_ptr
        own, gathered = self._factors
        R, N = gathered.shape[0], self.num_gaussians()
        K = 1 + self._params["features_rest"].shape[1]
        st = []
        for name in FEATURE_GROUPS:
            p = self._params[name]
            opt = self.optimizers[name]
            s_ = opt.state[p]
            if "exp_avg" not in s_:
                s_["step"] = torch.tensor(0.0)
                s_["exp_avg"] = torch.zeros_like(p)
                s_["exp_avg_sq"] = torch.zeros_like(p)
            s_["step"] += 1
            st.append((p.data, s_["exp_avg"], s_["exp_avg_sq"], float(opt.param_groups[0]["lr"])))
        # (the view directions of the step that produced the factors: the means have been stepped since)
        rc = load().fsgs_sh_coeff_grad_adam(R, N, K, sh_degree, ptr(self._means_prev), ptr(gathered), 1.0 / R,
                                            ptr(st[0][0]), ptr(st[0][1]), ptr(st[0][2]), st[0][3],
                                            ptr(st[1][0]), ptr(st[1][1]), ptr(st[1][2]), st[1][3],
                                            int(step_no), 0.9, 0.999, self.optim_cfg.eps, stream_ptr(self.device))
        if rc != 0:
            raise RuntimeError(f"fsgs_sh_coeff_grad_adam failed ({rc})")

    def _reduce_and_step(self, optimizer_step: bool) -> None:
        """Gradient exchange + Adam for the gradients now in the slab.  With several ranks (and an optimizer step)
        the feature half is only STARTED here; ``flush`` completes it when the next frame needs the colours."""
        assert self._pending is None, "the previous step's feature update has not been flushed"
        split = self._split_step(optimizer_step)
        if not split:
            self.sync_optimizer_state()  # (a full Adam step must not read moments that are stale outside the own shard)
            with self._comm():
                self.slab.all_reduce_mean_()
            if optimizer_step:
                self.optimizers["means"].param_groups[0]["lr"] = self._means_lr(self.step)
                self._optimizer_step(PARAM_ORDER)
            return
        sharded = self._geo is not None
        if not sharded:
            with self._comm():
                self.slab.all_reduce_geometry_mean_()
        factors = getattr(self, "_factors_used", None)
        self._factors_used = None
        if factors is not None:
            own, gathered = factors
            work = None
            if GradSlab._exchange():
                work = comm.all_gather_blocks_async(gathered, own)
                self.comm_bytes_last_step = 4 * (self.slab.split * 2 + gathered.numel())
            deg = self._sh_degree_now()
            mp = getattr(self, "_means_prev", None)
            if mp is None or mp.shape != self._params["means"].shape:
                mp = self._means_prev = torch.empty_like(self._params["means"].data)
            mp.copy_(self._params["means"].data)  # before Adam moves them (12 B per Gaussian)

            def finish(step_no):
                if work is not None:
                    with self._comm():  # (what is left of the gather once the next frame's binning + sort are enqueued)
                        work.wait()
                self._features_from_factors(deg, step_no)  # (rebuild + Adam in one launch)
            finish.applies_update = True
        else:
            finish = self.slab.all_reduce_features_async()
        self.optimizers["means"].param_groups[0]["lr"] = self._means_lr(self.step)
        step_no = self._sharded_geometry_step() if sharded else self._optimizer_step(GEOMETRY_GROUPS)
        self._pending = (finish, step_no)

    def _adam_in_backward_ok(self, optimizer_step: bool) -> bool:
        """One rank, the library's own Adam on all six groups right after the backward, nothing else reading the
        gradients (no exchange, no deferred / riding feature update, no sharded geometry step)."""
        mode = self.adam_in_backward_mode
        if not optimizer_step or mode == "0" or (mode != "1" and self.num_gaussians() < self.adam_in_backward_min_n):
            return False
        return (self._one_rank_fused() and not self._split_step(optimizer_step) and self._geo is None
                and self._pending is None and not GradSlab._exchange())

    def _grads_written(self) -> None:
        """This step leaves its gradients in the slab: the parameters' ``.grad`` are its views again."""
        self.last_step_grads_valid = True
        for name in PARAM_ORDER:
            p = self._params[name]
            if p.grad is None:
                p.grad = self.slab.views[name]

    def gradients(self) -> Dict[str, Tensor]:
        """The last step's parameter gradients (views of the slab, averaged over the ranks once the step has reduced
        them).  Raises when the last step applied its Adam update inside the backward launch: such a
        step writes no gradients — ``.grad`` of every parameter is None after it and ``last_step_grads_valid`` False —
        so hooks, gradient-norm logging or clipping must either run the step with ``adam_in_backward_mode = "0"`` or
        ask for gradients explicitly (``train_step(..., optimizer_step=False)``)."""
        if not self.last_step_grads_valid:
            raise RuntimeError("the last train_step applied Adam inside the backward launch and wrote no gradients "
                               "(set trainer.adam_in_backward_mode = '0' or FSGS_ADAM_IN_BACKWARD=0 to keep them)")
        return self.slab.views

    @staticmethod
    def _view_key(camera: Camera):
        """What identifies a VIEW (pose + intrinsics + size): per-view state — occlusion cuts, bucket room — is kept
        under it; computed once per camera object."""
        key = camera.__dict__.get("_view_key")
        if key is None:
            key = camera.__dict__["_view_key"] = (camera.c2w.to(torch.float32).cpu().numpy().tobytes(), camera.fx,
                                                  camera.fy, camera.cx, camera.cy, camera.width, camera.height)
        return key

    def _view_cuts(self, camera: Camera):
        """(key, cuts this view was last rendered with or None, buffer for this frame's cuts) — or (None, None, None)
        when occlusion cuts are off for this model size."""
        mode = self.occlusion_cut_mode
        if mode == "0" or (mode != "1" and self.num_gaussians() < self.occlusion_cut_min_n):
            return None, None, None
        key = self._view_key(camera)
        tiles = math.ceil(camera.width / 16) * math.ceil(camera.height / 16)
        zin = self._zcuts.get(key)
        if zin is not None and zin.numel() != tiles:
            zin = None
        return key, zin, torch.empty(tiles, dtype=torch.float32, device=self.device)

    def drop_occlusion_cuts(self) -> None:
        """Forgets every view's cuts (after an opacity reset the saturation depths of the previous frames say nothing
        about the next ones; keeping them would only cost second looks and redone frames)."""
        self._zcuts.clear()
        self._zcut_widen.clear()

    def flush(self) -> None:
        """Completes a deferred feature update (no-op otherwise).  Called before the colours are evaluated, before
        densification touches the parameters, and by anything that reads them (checkpoints, exports)."""
        if self._pending is None:
            return
        finish, step_no = self._pending
        self._pending = None
        if getattr(finish, "applies_update", False):
            finish(step_no)
            return
        if finish is not None:
            with self._comm():
                finish()
        self._optimizer_step(FEATURE_GROUPS, step_no)

    # -- the step as one library call (express.py) ---------------------------------------------------------------------
    def _express_ok(self, camera: Camera, target, is_fb: bool, factors, zin) -> bool:
        """The conditions of fused._FusedGetOutputs' count-free route (``use_fill``) + the benchmark loss + nothing
        pending: everything else keeps the per-op route."""
        from . import fused, ops
        if not self.express or factors is not None or zin is not None or self._pending is not None:
            return False
        if is_fb:
            # get_loss_dict: the terms ops._FusionLoss evaluates for FusionSense's configuration; a second depth target,
            # the cosine term and normals from depth keep the per-op route
            cfg, fb = self.loss_cfg, target
            if (ops.mono_depth_weights(cfg, fb) is not None or getattr(cfg, "normal_supervision", "mono") == "depth"
                    or (cfg.use_normal_loss and getattr(cfg, "use_normal_cosine_loss", False))):
                return False
            for t in (fb.image, fb.sensor_depth, fb.normal, fb.mask):
                if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous()):
                    return False
            if fb.width != camera.width or fb.height != camera.height:
                return False
        if self.half_attributes or not (fused.BIN_FILL and fused.SH_RIDES_WITH_SCAN and fused.ONE_LAUNCH_GAUSSIAN_BWD
                                        and fused.IMAGE_GRADS_IN_BWD and fused.FWD_TILE_ORDER and ops.USE_BIN_LIVE
                                        and ops.BIN_FILL_SORT_ORDER and ops.FILL_WAIT_ON_FLAG and ops.DEFER_COMBINE):
            return False
        N = self.num_gaussians()
        if not (0 < N < min(fused.BIN_FILL_MAX_N, fused.KEPT_MIN_N)) or self._params["features_rest"].shape[1] != 15:
            return False
        if self._sh_degree_now() > 3 or fused.FWD_WALK.choice(N)[0] != fused.FWD_WALK.ONE_WAVE:
            return False
        tiles = math.ceil(camera.width / 16) * math.ceil(camera.height / 16)
        if tiles > self._bin_live_max_tiles() or ops.bin_live_is_dense(self.device, N, tiles):
            return False
        if is_fb:
            return True
        for k in ("rgb", "depth"):
            t = target.get(k)
            if t is None or not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                return False
        tn = target.get("normal")
        return tn is None or (tn.is_cuda and tn.dtype == torch.float32 and tn.is_contiguous())

    def _bin_live_max_tiles(self) -> int:
        v = getattr(self, "_max_tiles", None)
        if v is None:
            from ._lib import load
            v = self._max_tiles = int(load().fsgs_bin_live_max_tiles())
        return v

    def _express_step(self, camera: Camera, cap: int, has_normal: bool):
        """The argument blocks of this frame shape (re-made when the model, the capacity or the shape has changed)."""
        from .express import ExpressStep
        ex = self._express
        # (the estimate `cap` moves with every new maximum of the frame shape: the blocks are sized with some headroom
        # and kept while the estimate stays between half of and all of their capacity — a larger capacity only makes an
        # overflow less likely, the lists are the same)
        if (ex is None or ex.key[:3] != (self.num_gaussians(), camera.width, camera.height) or ex.key[4] != bool(has_normal)
                or not (ex.capacity // 2 <= cap <= ex.capacity) or not ex.still_fits(self)):
            ex = self._express = None  # (free the old buffers first)
            roomy = -(-int(cap * 1.125) // 65536) * 65536
            ex = self._express = ExpressStep(self, camera.width, camera.height, roomy, has_normal)
        return ex

    def train_step(self, camera: Camera, target: Dict[str, Tensor], optimizer_step: bool = True):
        if getattr(self, "_one", None) is None or self._one.device != self.device:
            self._one = torch.ones((), dtype=torch.float32, device=self.device)
        from .losses import FrameBatch
        is_fb = isinstance(target, FrameBatch)
        if self.strategy is not None and hasattr(self.strategy, "before_train"):
            self.strategy.before_train(self)
        direct = self.fused and self.direct and self.device.type == "cuda" and (is_fb or "depth" in target)
        if not direct:
            # the previous step's deferred feature update reads the slab: it must land before the slab is cleared
            # (the tape-free step hands flush to the frame instead, which calls it right before the SH colours)
            self.flush()
        if direct:
            from .fused import fused_step_forward_backward
            stats, add_mask, bthr = self._frame_state(camera, True)
            # (the kernels read the anchors as a uint8 row: the strategy keeps the converted mask with the mask)
            frozen = (self.strategy.add_mask_u8() if (add_mask is not None and hasattr(self.strategy, "add_mask_u8"))
                      else add_mask)
            factors = self._factor_buffers(optimizer_step)
            from .ops import LiveListOverflow
            from .ops import OcclusionCutInvalid
            view_key, zin, zout = self._view_cuts(camera)
            # count-free binning (fused.BIN_FILL): the room every tile's bucket needs comes from this view's previous
            # frame.  The buckets belong to TILES, not to Gaussians: a refinement that moves the model's size by a few
            # per cent moves the tiles' lists by as much (inside the buckets' growth room; a tile that does outgrow its
            # bucket redoes the frame); a model that changed by more than bin_hist_keep_rel starts its views over
            bin_key = self._view_key(camera) if (self.no_wait and self.num_gaussians() > 0) else None
            if self._bin_hist_n != self.num_gaussians():
                if abs(self.num_gaussians() - self._bin_hist_n) > self.bin_hist_keep_rel * max(self._bin_hist_n, 1):
                    self._bin_hist.clear()
                self._bin_hist_n = self.num_gaussians()
            bin_hist = self._bin_hist.get(bin_key) if bin_key is not None else None
            # (frames binned with occlusion cuts hold a fraction of the pairs: their own capacity estimate, or the first
            # uncut frame after them — a new view, a redone frame — would overflow it)
            cap_key = self._live_caps.key(self.device, 1, self.num_gaussians(), camera.width, camera.height,
                                          zin is not None)
            cap = self._live_caps.get(cap_key) if (self.no_wait and self.num_gaussians() > 0) else 0
            aib, aib_step = None, None
            aib_ok = self._adam_in_backward_ok(optimizer_step)
            if aib_ok:
                self.optimizers["means"].param_groups[0]["lr"] = self._means_lr(self.step)
                aib_step = getattr(self, "adam_steps", 0) + 1
            # (the one-call route: with the Adam step inside the backward, or — optimizer_step=False on one rank — writing
            # the gradients; a step whose optimizer runs as launches of its own keeps the per-op route)
            grad_mode = (not optimizer_step) and self._one_rank_fused() and self._geo is None
            use_express = ((aib_ok or grad_mode) and bin_hist is not None and cap > 0
                           and self._express_ok(camera, target, is_fb, factors, zin))
            if aib_ok and not use_express:
                aib = self._fused_adam_step(PARAM_ORDER, aib_step, as_groups=True)
            if zin is not None:
                self.cut_frames += 1
            # The step as ONE library call (express.py; fsgs_step_run) where the count-free route applies: same launches,
            # same arguments, same results — the host issues the step in microseconds instead of ~0.45 ms.
            express_done = False
            if use_express:
                from .ops import bin_fill_words
                ex = self._express_step(camera, cap, True if is_fb else "normal" in target)
                if bin_fill_words(int(bin_hist["n_live"]), ex.T) > ex.words:  # (a view last seen under a larger model)
                    use_express = False
                    if aib_ok:
                        aib = self._fused_adam_step(PARAM_ORDER, aib_step, as_groups=True)
            if use_express:
                if aib_ok:
                    ex.adam.step = int(aib_step)
                    for k, name in enumerate(PARAM_ORDER):
                        ex.adam.lr[k] = float(self.optimizers[name].param_groups[0]["lr"])
                over, n_live_x, loss, out = ex.run(self, camera, target, bin_hist, stats, frozen, bthr,
                                                   ex.adam if aib_ok else None,
                                                   self._sh_degree_now(), self._one,
                                                   fusion=((self.loss_cfg,) + self._touch_rows()) if is_fb else None)
                if over:  # the frame outgrew its lists / a bucket: once more through the exact two-pass route (below)
                    self.live_overflows += 1
                    self._bin_hist.pop(bin_key, None)
                    bin_hist = None
                    self._live_caps.raise_to(cap_key, n_live_x)
                    cap = 0
                    bthr = None  # (launch 1 has written the binary opacities: the redo must not threshold them again)
                    if aib_ok:
                        aib = self._fused_adam_step(PARAM_ORDER, aib_step, as_groups=True)
                else:
                    express_done = True
                    self.express_steps += 1
            for attempt in (() if express_done else (0, 1, 2)):
                try:
                    loss, out = fused_step_forward_backward(
                        self._params, camera, target, self._sh_degree_now(), self.device, self.slab.views, self._one,
                        stats_out=stats, add_mask=frozen, binary_threshold=bthr,
                        pre_sh=self.flush if self._pending is not None else None, adam_rider=self._adam_rider(),
                        sh_factors_out=factors[0] if factors else None,
                        fusion=((self.loss_cfg,) + self._touch_rows()) if is_fb else None,
                        half=self.half_mirrors(), live_capacity=cap, zcut_in=zin, zcut_out=zout,
                        zcut_margins=self.zcut_margins if view_key not in self._zcut_widen else tuple(
                            self._zcut_widen[view_key] * x for x in self.zcut_margins[:2]) + self.zcut_margins[2:],
                        adam_in_backward=aib, bin_hist=bin_hist if cap > 0 else None,
                        want_bin_hist=bin_key is not None)
                    break
                except OcclusionCutInvalid:  # a cut tile did not saturate inside its prefix: the exact frame, uncut
                    self.cut_redone += 1
                    self._zcut_widen[view_key] = min(8.0, 2.0 * self._zcut_widen.get(view_key, 1.0))
                    zin = None
                    zout = torch.empty_like(zout)
                    bthr = None  # (as below: the abandoned attempt's count pass has run)
                    cap_key = cap_key[:-1] + (False,)
                    cap = self._live_caps.get(cap_key) if (self.no_wait and cap > 0) else 0
                except LiveListOverflow as e:  # rare: the frame outgrew the estimate -> once more, with exact sizes
                    self.live_overflows += 1
                    self._bin_hist.pop(bin_key, None)  # (or a tile outgrew its bucket: the redone frame leaves new room)
                    bin_hist = None
                    self._live_caps.raise_to(cap_key, e.needed)
                    cap = 0
                    # the abandoned attempt's count pass has already written the binary opacities (and landed any
                    # pending feature update): the logits now hold the written values, the retry must not
                    # threshold them a second time
                    bthr = None
                    if zout is not None:
                        zout = torch.empty_like(zout)  # (the truncated frame's cuts are not kept)
            else:  # (cannot happen: the third attempt runs uncut and with exact sizes)
                if not express_done:
                    raise RuntimeError("a frame was abandoned three times (occlusion cut / live-list capacity)")
            if view_key is not None:
                self._zcuts.pop(view_key, None)
                if out["info"].zcut_out is not None:  # (None: the frame took a binning route without cuts)
                    self._zcuts[view_key] = zout  # (most recently used last)
                    while len(self._zcuts) > self.zcut_max_views:  # e.g. poses that move every step never repeat
                        old_key = next(iter(self._zcuts))
                        del self._zcuts[old_key]
                        self._zcut_widen.pop(old_key, None)
            n_live = out["info"].n_live
            if n_live is not None and self.no_wait:
                self._live_caps.update(cap_key, n_live)
            self.bin_frames[out["info"].bin_route] = self.bin_frames.get(out["info"].bin_route, 0) + 1
            hist_out = out["info"].bin_hist_out
            if bin_key is not None:
                self._bin_hist.pop(bin_key, None)
                if hist_out is not None and n_live is not None:
                    hist_out["n_live"] = int(n_live)
                    self._bin_hist[bin_key] = hist_out  # (most recently used last)
                    while len(self._bin_hist) > self.zcut_max_views:
                        del self._bin_hist[next(iter(self._bin_hist))]
            self._factors_used = factors
            if (aib is not None or express_done) and out["info"].adam_applied:
                # the backward launch has stepped all six groups: count the step, nothing left to launch
                self.adam_steps = aib_step
                for name in PARAM_ORDER:
                    self.optimizers[name].state[self._params[name]]["step"] += 1
                self.adam_in_backward_steps += 1
                optimizer_step = False
                # no gradient was written: nothing may read the previous step's numbers through .grad / the slab
                self.last_step_grads_valid = False
                for name in PARAM_ORDER:
                    self._params[name].grad = None
            else:
                self._grads_written()
        else:
            if not self.fused:
                self.slab.zero_()  # autograd accumulates into the slab views; the fused node overwrites them
            out = self.forward(camera)
            loss = self.loss(out, target, camera)
            self._grads_written()
            loss.backward(gradient=self._one)  # (the default would launch a fill kernel for the seed gradient)
            if is_fb and self.fused:
                from .ops import fusion_loss_weights, min_scale_grad_
                g_min = fusion_loss_weights(self.loss_cfg, target, self.num_gaussians(), 0)[2]
                if g_min != 0.0:
                    min_scale_grad_(self._params["scales"].data, g_min, self._one.reshape(1), self.slab.views["scales"])
        self._reduce_and_step(optimizer_step)
        if self.strategy is not None:
            self.strategy.after_train(self, out, camera)
            self.strategy.maybe_refine(self)
        self.step += 1
        self.last_info = out["info"]
        return loss.detach(), out
