"""``DNSplatterModel.get_outputs`` as ONE autograd node.

Semantically identical to :func:`fusionsense_amd.fusion.render_fusionsense` — which mirrors the
reference's op-by-op formulation (/root/reference/dn_splatter/dn_model.py:469-671) through the
drop-in ``rasterization`` / ``rasterize_gaussians`` surface — but organised the way the hardware
wants it:

* the RGB+depth pass and the normal pass see the same centres, conics, opacities and sorted
  lists, so they are composited in ONE walk over 64-byte live-list records (4 + 3 channels; the
  normal plane's gradient is kept away from the projected centres and ``absgrad`` exactly as the
  reference's ``xys.detach()`` does);
* activations, SH evaluation straight from the split ``features_dc`` / ``features_rest``
  storage, the rgb / depth / normal epilogue and all their VJPs are single kernels; no
  ``torch.cat`` of the 192 B/Gaussian SH tensor, no per-op autograd nodes, no gradient
  accumulation kernels;
* gradients are written straight into views of the trainer's flat gradient slab.

Everything goes through libfsgs.so; results are tested against the unfused path and the oracle.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import torch
from torch import Tensor

from . import ops
from ._lib import load, ptr, stream_ptr
from .ops import WORKSPACE, _run, tile_bits
from .scenes import Camera

TILE = 16


class FrameInfo:
    """Per-frame side outputs (the ``meta`` of gsplat + what after_train reads)."""

    def __init__(self):
        self.radii = self.means2d = self.depths = self.conics = self.tiles_per_gauss = None
        self.isect_ids = self.payload = self.isect_offsets = self.last_ids = None
        self.absgrad: Optional[Tensor] = None  # [1,N,2], set by backward
        self.normals_world = None
        self.legacy_rule_diff = 0
        # after_train statistics fused into the backward: {"xys_grad_norm","vis_counts","max_2Dsize": [N] tensors,
        # "inv_max_hw": float} supplied by the caller; stats_done tells it they have been applied
        self.stats_out: Optional[dict] = None
        self.stats_done = False
        self.frozen: Optional[Tensor] = None           # [N] uint8: touch anchors (dn_model.py:535-541)
        self.binary_threshold: Optional[float] = None  # perform the binary-opacity write (dn_model.py:492-503)
        # called right before the SH colours are evaluated (the trainer's deferred feature update, data-parallel runs)
        self.pre_sh = None
        # a deferred Adam step that may ride in the binning's count launch: an object with `.groups` (ops.adam_groups)
        # and `.consumed()` (called once the launch that carries it has been enqueued); `pre_sh` stays the fallback
        self.adam_rider = None
        # [N + 1,4] buffer: the backward writes the factors of the SH coefficient gradients there INSTEAD of the
        # gradients themselves (trainer's factored exchange, data-parallel runs)
        self.sh_factors_out: Optional[Tensor] = None
        # {"depth","normal" (or None),"seed","g_depth","g_normal"}: evaluate the depth / normal L1 terms with the
        # images (fsgs_epilogue_loss_fwd); the results land in aux_loss = (partial sums, v_depth, v_normal)
        self.loss_targets: Optional[dict] = None
        self.aux_loss = None
        # BASELINE config #5: IEEE-half mirrors of scales / quats / features_dc / features_rest / opacities that the
        # render kernels read instead of the fp32 masters (the trainer keeps them in step with the parameters)
        self.half: Optional[Dict[str, Tensor]] = None
        # no-wait binning (the trainer's tape-free step): list buffers are sized for this many live pairs instead of
        # waiting for the frame's count; ``pending_count`` is then checked by the caller once the forward's launches
        # are enqueued (ops.bin_live_check), ``n_live`` is the true total after that check
        self.live_capacity = 0
        # count-free binning (ops.project_bin_live_fill_async): the view's bucket bases from its previous frame
        # ({"base": int32 [T + 1], "n_live": pairs of that frame} or None), whether this frame should leave the bases for
        # the next one, and where it leaves them ({"base": ...}; the caller adds "n_live" once it knows it)
        self.bin_hist_in: Optional[dict] = None
        self.bin_hist_want = False
        self.bin_hist_out: Optional[dict] = None
        self.bin_route = "two_pass"  # or "fill": the count-free route
        self.pending_count = None
        self.n_live: Optional[int] = None
        # the forward compositing's walk this frame used (fsgs_raster_fwd_quad's ``walk``; FWD_WALK)
        self.fwd_walk = 0
        # occlusion cuts (dense scenes): ``zcut_in`` [T] float32 = per-tile depth behind which pairs
        # are not binned (the previous frame of this view saturated in front of it; +inf = no cut), ``zcut_out`` [T]
        # receives this frame's cuts for the next one.  A cut tile that does NOT saturate inside its prefix makes the
        # frame invalid (ops.OcclusionCutInvalid from the caller's check; to be redone without cuts).
        self.zcut_in: Optional[Tensor] = None
        self.zcut_out: Optional[Tensor] = None
        self.zcut_margins = (1.0, 0.03, 0.25)  # (span, rel, tail: fsgs_tile_zcut_update)
        self.pending_verdict = None
        self.kept: Optional[Tensor] = None  # [N] uint8 (dense scenes): the Gaussian is in at least one tile's list
        # an _lib.AdamGroups of the six parameter groups (trainer's order): the backward applies this Adam step in its
        # per-Gaussian launch instead of writing gradients (fsgs_gauss_sh_bwd_adam); adam_applied tells the caller
        self.adam_in_backward = None
        self.adam_applied = False
        self.min_scale_g = 0.0  # (with adam_in_backward: the min-scale term's gradient weight, folded into that launch)

    @property
    def flatten_ids(self):
        """Gaussian id of every LIVE list entry (the sorted payload carries the quadrant mask in its top 4 bits)."""
        return None if self.payload is None else self.payload & 0x0FFFFFFF

    def __getitem__(self, key):  # dict-style access like gsplat's meta
        return getattr(self, key)


def _camera_on_device(camera: Camera, dev) -> Dict[str, Tensor]:
    """viewmat / K / c2w / campos as device tensors, computed once per (camera, device) on the host
    (get_viewmat of dn_model.py:550 is a handful of 4x4 operations)."""
    cache = camera.__dict__.setdefault("_dev_cache", {})
    key = str(dev)
    if key not in cache:
        c2w = camera.c2w.to(torch.float32)
        R = c2w[:3, :3] * torch.tensor([[1.0, -1.0, -1.0]])
        Rinv = R.T
        viewmat = torch.eye(4)
        viewmat[:3, :3] = Rinv
        viewmat[:3, 3] = -(Rinv @ c2w[:3, 3])
        cache[key] = dict(viewmat=viewmat[None].contiguous().to(dev), K=camera.K()[None].contiguous().to(dev),
                          c2w=c2w.contiguous().to(dev), campos=c2w[:3, 3][None].contiguous().to(dev))
    return cache[key]


_ACCUM: Dict[str, Tensor] = {}
_ONES3: Dict[str, Tensor] = {}


# SH forward + packing in the launch of the binning's table scan (fsgs_project_bin_live_count_sh_pack); 0 = own launch
SH_RIDES_WITH_SCAN = os.environ.get("FSGS_SH_RIDES_WITH_SCAN", "1") != "0"
# SH backward + per-Gaussian backward in one launch (fsgs_gauss_sh_bwd) where it applies; 0 = the two launches
ONE_LAUNCH_GAUSSIAN_BWD = os.environ.get("FSGS_ONE_LAUNCH_GAUSSIAN_BWD", "1") != "0"
GRAD_REPLICAS = 0 if os.environ.get("FSGS_GRAD_REPLICAS", "1") != "0" else 1  # 0: ask the library (fsgs_grad_replica_lines)
# Scenes above this size keep one line per Gaussian: the two readers pay for three more lines per spread Gaussian,
# and at 6-10 M Gaussians (configs #4 / #5, where ~18 % of the visible ones pass the footprint threshold) that costs
# more than the compositing backward gains (measured: +0.11 / +0.17 ms per step with replicas).
GRAD_REPLICAS_MAX_N = int(os.environ.get("FSGS_GRAD_REPLICAS_MAX_N", str(1 << 20)))


def _grad_lines(N: int) -> int:
    global GRAD_REPLICAS
    if GRAD_REPLICAS == 0:
        from ._lib import load
        GRAD_REPLICAS = int(load().fsgs_grad_replica_lines())  # (4 lines by footprint tier 1, all of them by tier 2)
    return GRAD_REPLICAS if N <= GRAD_REPLICAS_MAX_N else 1


def _grad_accumulator(dev, N: int) -> Tensor:
    """The packed gradient accumulator of the backward, [_grad_lines(N) * N, 16]: line n plus, for Gaussians with a
    large footprint, three replica lines N rows apart (csrc/common.h: grad_spread).  Kept zeroed between frames by
    its reader (one per device; re-created when N changes, i.e. after densify / prune)."""
    key = str(dev)
    t = _ACCUM.get(key)
    if t is None or t.shape[0] != _grad_lines(N) * N:
        t = torch.zeros(_grad_lines(N) * N, 16, dtype=torch.float32, device=dev)
        _ACCUM[key] = t
        if os.environ.get("FSGS_DEBUG_PTRS"):
            import sys
            print(f"[fsgs] grad accumulator N={N} at 0x{t.data_ptr():x}", file=sys.stderr)
    return t


IMAGE_GRADS_IN_BWD = os.environ.get("FSGS_IMAGE_GRADS_IN_BWD", "1") != "0"
# queue positions for the backward's extra workgroups (groups of 8 segments of streams longer than 8 segments: fsgs.h,
# seg_split / bwd_queue); 0 = every quadrant's stream is walked by its own workgroup alone (round 4)
BWD_QUEUE_ITEMS = 1024
# the backward's quadrants start longest own share first (fsgs.h: bwd_order; needs the queue); classes = the library's
BWD_ORDER = True
BWD_ORDER_CLASSES = 32
# the forward walks the tiles longest list first where the binning left that order (count-free route: ops.BIN_FILL_SORT_ORDER)
FWD_TILE_ORDER = True
BWD_ORDER_DENSE = False  # (models of 2^20 Gaussians and more — no queue items there: measured, config #4 0.939 ms either way)
# count-free binning of revisited views (ops.project_bin_live_fill_async; fsgs.h): on / off, and the model size from which
# the dense machinery (occlusion cuts, several chunks per workgroup) keeps the two-pass route
BIN_FILL = True
BIN_FILL_MAX_N = 1 << 20
# from this many Gaussians on, the per-Gaussian backward fetches SH coefficients lazily / skips idle Adam elements
# (flags of fsgs_gauss_sh_bwd*, decided here per launch; module attributes so that tests can move them)
LAZY_SH_MIN_N = int(os.environ.get("FSGS_LAZY_SH_MIN_N", str(1 << 20)))
SKIP_IDLE_MIN_N = int(os.environ.get("FSGS_SKIP_IDLE_MIN_N", str(1 << 20)))
# from this many Gaussians on, the SH forward evaluates only Gaussians that are in some tile's list (fsgs.h: kept)
KEPT_MIN_N = int(os.environ.get("FSGS_KEPT_MIN_N", str(1 << 20)))


class _BwdDispatch:
    """The dispatch order of the compositing backward (the ``dispatch_stride`` argument of fsgs_raster_bwd_quad*):
    quadrants in row-major order, or consecutive workgroups ``stride`` quadrants apart.  A RULE of the frame's size since
    round 5 (rounds 3-4 timed both orders on a few early frames of every frame shape and kept the faster one: two runs of
    a seed could differ, ranks could disagree, and the duplicate launches needed a scratch accumulator): frames of up to
    ``small_frame_pixels`` pixels — fewer quadrants than a few rounds of resident workgroups, where neighbours' equally
    long lists would otherwise end up side by side on a CU — are dispatched 7 apart (config #2: 0.169 vs 0.176 ms;
    config #3: 0.208 vs 0.211), larger ones row-major (config #4: 0.927 vs 0.947).  ``forced`` / ``forced_stride``: tests."""
    small_frame_pixels = 1 << 20
    small_frame_stride = 7

    def __init__(self):
        self.forced = False
        self.forced_stride = 0

    def stride(self, W: int, H: int) -> int:
        if self.forced:
            return self.forced_stride
        return self.small_frame_stride if W * H <= self.small_frame_pixels else 0


BWD_DISPATCH = _BwdDispatch()


class _FwdWalk:
    """How the fused node's forward compositing walks the lists (the ``walk`` / ``handoff_records`` / ``tail_items``
    arguments of fsgs_raster_fwd_quad).  ONE walk for every frame since round 5, so two runs of a seed give the same
    frames bit for bit: one wave per 8x8 quadrant (27 vector instructions per record), and a wave that has streamed
    ``handoff_records`` records of a list far longer than the frame's mean, with pixels still open, hands the rest of the
    list to the tail waves of the same launch, which finish it chunk-parallel (raster_quad.hip: TailQueue) — the frame's
    longest list no longer bounds the launch.  Rounds 3-4 chose between this walk WITHOUT the hand-off and the
    four-waves-per-quadrant walk by timing them per frame shape (config #2: 79 vs 95 us; config #3's trained scene with
    lists of 2 800 entries: 278 vs 153 us); that tuner is gone.

    What IS still decided from what earlier frames showed is only how many tail workgroups a launch carries (``tail_items``:
    0 while no list of this device's frames has asked for a hand-off lately — then waiting tail workgroups would cost
    ~4 us per launch for nothing, config #2 — else ``max_items``).  It cannot change a result: a list that finds no tail
    is finished by its own wave with the chunks' arithmetic, the same bits (tests: a queue of 8 positions).  The demand
    is a counter in the scratch, copied to pinned memory every ``PROBE_EVERY`` frames without a host wait.

    ``forced`` / ``forced_walk`` / ``handoff_records`` are attributes so that tests and bench.py can run the other walks
    (0 = four waves per quadrant; 1 = one wave; handoff_records = 0: never hand off); forced walks always carry the tail."""
    ONE_WAVE, FOUR_WAVES = 1, 0
    PROBE_EVERY, KEEP_FRAMES = 16, 1024

    def __init__(self):
        self.walk = self.ONE_WAVE
        # records a wave streams before it hands the rest of its list off (0 = never), if the list is more than
        # ``handoff_rel_len`` times the frame's mean list length.  Measured (MI355X, us per launch, config #2 / #3's
        # schedule window): no hand-off 78 / 255-274; 64 / 128 / 192 records at 4 x the mean: 81 / 82 / 82 and
        # 104 / 102 / 101 (at 8 x: 126 at config #3); every list whatever its length: 101-122 / 99-104; the four-wave
        # walk: 96 / 152.  The ~4 us at config #2 are tail workgroups waiting for the base waves to leave: not launched
        # while nothing asks for them (see above).
        self.handoff_records = 192
        self.handoff_rel_len = 4
        # ... and only in frames in which SOME list is more than this many times the mean (0 = any frame): a frame
        # statistic the in-tile sort leaves on the device.  Config #2's longest list is 5.7 x its mean in most frames (a
        # few reach 8 x): next to nothing is queued there — its quadrants saturate early, queueing cost 5 us per launch;
        # config #3's hull tiles: 16 x (gate 0 / 8 / 12: 102 / 101 / 101 us).
        self.handoff_gate_len = 12
        # dense scenes (the depth-slab binning route, configs #4 / #5): tens of thousands of quadrants keep every SIMD busy
        # and every list is long — nothing to gain, a queue to overflow: no hand-off from this many Gaussians on
        self.handoff_max_n = 1 << 20
        self.max_items = 16384       # queue positions = tail workgroups of a launch that carries the tail
        self.forced = False          # tests: use ``forced_walk`` (and ``handoff_records`` as set)
        self.forced_walk = 0
        self.failed = False          # a hand-off wait ever ran into its bound on some device (see scratch())
        self.state: Dict = {}        # device -> dict(scratch, epoch, frames, keep_until, seen, probe)

    def choice(self, n_gaussians: int = 0):
        walk = self.forced_walk if self.forced else self.walk
        on = walk == self.ONE_WAVE and (self.forced or n_gaussians < self.handoff_max_n)
        return walk, (self.handoff_records if on else 0)

    def _state(self, dev):
        key = str(dev)
        st = self.state.get(key)
        need = int(load().fsgs_raster_fwd_tail_scratch_bytes(int(self.max_items)))
        if st is None or st["scratch"].numel() != need:
            st = self.state[key] = dict(scratch=torch.zeros(need, dtype=torch.uint8, device=dev), epoch=0, frames=0,
                                        keep_until=0, seen=0, probed_at=0, probe=None,
                                        pinned=torch.zeros(2, dtype=torch.int32).pin_memory())
        return st

    def scratch(self, dev):
        """(tail scratch of this device — zeroed once —, tail workgroups of this launch, its epoch)."""
        st = self._state(dev)
        st["frames"] += 1
        if st["probe"] is not None and st["probe"].query():
            st["probe"] = None
            if int(st["pinned"][0]) != 0:
                # TailHeader.error: a bounded look-back wait of the hand-off ran into its bound (a broken dispatch-order
                # assumption: CU masking, a pre-empted or shared GPU, another driver) — the frames since the last probe
                # may hold wrong pixels and gradients.  Never silently: the hand-off is switched off for good and the
                # caller is told.  The word is sticky; the scratch is dropped so that a run that carries on
                # (after catching this) starts from a clean one, without hand-off.
                self.handoff_records = 0
                self.failed = True
                self.state.pop(str(dev), None)
                raise RuntimeError("fsgs: the forward hand-off protocol timed out on this device (TailHeader.error); frames "
                                   f"of the last {self.PROBE_EVERY} steps may be wrong — the hand-off is now disabled "
                                   "(FWD_WALK.handoff_records = 0)")
            demand = int(st["pinned"][1])
            # (at least one hand-off per frame since the last look: a stray long list now and then — config #2 has one
            # in every tenth frame — is finished by its own wave)
            if demand - st["seen"] >= st["frames"] - st["probed_at"]:
                st["keep_until"] = st["frames"] + self.KEEP_FRAMES
            st["seen"], st["probed_at"] = demand, st["frames"]
        if st["probe"] is None and st["frames"] % self.PROBE_EVERY == 0:
            # (TailHeader.error and .demand: the int32s at byte offsets 8 and 12 — a protocol timeout ever, and the lists
            # handed off, or wanting to be, so far)
            st["pinned"].copy_(st["scratch"][8:16].view(torch.int32), non_blocking=True)
            st["probe"] = torch.cuda.Event()
            st["probe"].record()
        items = int(self.max_items) if (self.forced or st["frames"] <= st["keep_until"]) else 0
        if items > 0:
            st["epoch"] += 1  # (two launches WITH a tail alternate the queue's counters)
        return st["scratch"], items, st["epoch"]

    def tail_items_now(self, dev) -> int:
        st = self.state.get(str(dev))
        return 0 if st is None else (int(self.max_items) if (self.forced or st["frames"] <= st["keep_until"]) else 0)

    def error(self, dev) -> int:
        """1 if a look-back wait of any frame on this device ever ran into its bound (tests), else 0."""
        st = self.state.get(str(dev))
        return 0 if st is None else int(load().fsgs_raster_fwd_tail_error(st["scratch"].data_ptr(), stream_ptr(dev)))


FWD_WALK = _FwdWalk()


class _FusedGetOutputs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, scales, quats, features_dc, features_rest, opacities, cam, width, height,
                sh_degree, background, info: FrameInfo, grad_out: Optional[Dict[str, Tensor]]):
        ctx.set_materialize_grads(False)  # unused outputs (accumulation) arrive as None, not as a zero image
        lib = load()
        dev = means.device
        sp = stream_ptr(dev)
        N = means.shape[0]
        K = 1 + features_rest.shape[1]
        W, H = width, height
        tw, th = math.ceil(W / TILE), math.ceil(H / TILE)
        f32 = dict(dtype=torch.float32, device=dev)
        means, scales, quats, features_dc, features_rest, opacities = (
            t.contiguous() for t in (means, scales, quats, features_dc, features_rest, opacities))

        scales_exp = torch.empty(N, 3, **f32)
        opac_sig = torch.empty(N, **f32)
        radii = torch.empty(1, N, dtype=torch.int32, device=dev)
        means2d = torch.empty(1, N, 2, **f32)
        depths = torch.empty(1, N, **f32)
        conics = torch.empty(1, N, 3, **f32)
        bthr = info.binary_threshold
        hm = info.half  # half attribute mirrors, or None
        # live emission: only (Gaussian, tile) pairs that can reach a pixel are binned and sorted; the
        # quadrant masks ride in the payload (gsplat's full lists are not an output of get_outputs)
        opac_row = opac_sig.view(1, N)
        direct_bins = (ops.USE_BIN_LIVE and tw * th <= lib.fsgs_bin_live_max_tiles()
                       and not ops.bin_live_is_dense(dev, N, tw * th))
        normals_world = torch.empty(N, 3, **f32)
        packed = torch.empty(N, 16, **f32)
        n_cells = lib.fsgs_raster_quad_max_cells()
        # partial image maxima: zeroed by the pack kernel, raised by the forward — and behind them, zeroed by the same
        # kernel, the queue of the backward's extra workgroups (a counter + BWD_QUEUE_ITEMS int4 items: fsgs.h, seg_split)
        # ... behind the items the BWD_ORDER_CLASSES counters of the backward's dispatch order (fsgs.h: bwd_order)
        # (dense scenes, N >= 2^20: no queue items — every quadrant's stream is long and every SIMD busy: nothing gained —
        # but the dispatch order is kept: a queue of zero items is its counter, padding and the order's class counters)
        n_q_items = BWD_QUEUE_ITEMS if N < FWD_WALK.handoff_max_n else (0 if BWD_ORDER_DENSE else -1)
        q_words = 4 + 4 * max(n_q_items, 0) + BWD_ORDER_CLASSES
        max_last = torch.empty(n_cells + q_words + 4, **f32)
        # (dense scenes, N >= 2^20: every quadrant's stream is long and every SIMD busy — measured: nothing gained)
        bwd_queue = (max_last[n_cells:n_cells + q_words].view(torch.int32)
                     if (BWD_QUEUE_ITEMS > 0 and n_q_items >= 0) else None)
        # ... and the frame statistic that gates the forward's hand-off (set by the in-tile sort: fsgs_bin_live_emit)
        long_flag = max_last[n_cells + q_words:].view(torch.int32)
        # the SH colours + packing ride in the binning's scan launch when nothing stands between them and the
        # features (no deferred feature exchange) and the storage is the 16-coefficient split one
        adam_rides = direct_bins and info.adam_rider is not None and N > 0
        pre_sh = None if adam_rides else info.pre_sh  # (a riding update IS the pending feature update)
        sh_rides = direct_bins and SH_RIDES_WITH_SCAN and pre_sh is None and K == 16 and sh_degree <= 3 and N > 0
        if direct_bins:
            # pairs go straight into their tile's bucket (no emission-order lists), and the count pass projects the
            # Gaussians itself (activations + binary-opacity write included): no projection launch
            rider = None
            # dense scenes: the count pass notes which Gaussians are in any list at all; the riding SH forward skips
            # the others (no coefficient read, no record written — nothing gathers it)
            kept = torch.empty(N, dtype=torch.uint8, device=dev) if (sh_rides and N >= KEPT_MIN_N) else None
            info.kept = kept
            if sh_rides:
                rider = (sh_degree, cam["campos"], hm["features_dc"] if hm is not None else features_dc,
                         hm["features_rest"] if hm is not None else features_rest, cam["c2w"], packed, normals_world,
                         max_last)
            proj_out = dict(scales_exp=scales_exp, opac_sig=opac_sig, radii=radii, means2d=means2d, depths=depths,
                            conics=conics)
            hist = info.bin_hist_in
            use_fill = (BIN_FILL and sh_rides and hist is not None and int(info.live_capacity) > 0 and hm is None
                        and kept is None and info.zcut_in is None and not adam_rides and N < BIN_FILL_MAX_N
                        and hist["base"].numel() == tw * th + 1)
            if use_fill:
                # ONE enumeration: the pairs go straight into buckets sized from this view's previous frame
                count = ops.project_bin_live_fill_async(means, quats, scales, opacities, bthr, cam["viewmat"], cam["K"],
                                                        W, H, tw, th, proj_out, int(info.live_capacity), rider, hist)
            else:
                count = ops.project_bin_live_count_async(
                    means, quats, scales, opacities, bthr, cam["viewmat"], cam["K"], W, H, tw, th, proj_out, half=hm,
                    capacity=int(info.live_capacity), sh_pack=rider,
                    adam=info.adam_rider.groups if adam_rides else None, zcut=info.zcut_in, kept=kept)
            if adam_rides:
                info.adam_rider.consumed()
                info.adam_rider = None
        elif hm is not None:
            _run(lib.fsgs_project_fwd_act_h16, (1, N, ptr(means), ptr(hm["quats"]), ptr(hm["scales"]),
                                               ptr(hm["opacities"]), ptr(opacities), 0 if bthr is None else 1,
                                               0.0 if bthr is None else float(bthr), ptr(cam["viewmat"]),
                                               ptr(cam["K"]), W, H, 0.3, 0.01, 1e10, 0.0, ptr(scales_exp),
                                               ptr(opac_sig), ptr(radii), ptr(means2d), ptr(depths), ptr(conics), sp),
                 "fsgs_project_fwd_act")
        else:
            _run(lib.fsgs_project_fwd_act, (1, N, ptr(means), ptr(quats), ptr(scales), ptr(opacities),
                                           0 if bthr is None else 1, 0.0 if bthr is None else float(bthr),
                                           ptr(cam["viewmat"]),
                                           ptr(cam["K"]), W, H, 0.3, 0.01, 1e10, 0.0, ptr(scales_exp), ptr(opac_sig),
                                           ptr(radii), ptr(means2d), ptr(depths), ptr(conics), sp),
                 "fsgs_project_fwd_act")
            count = ops.isect_count_live_async(means2d, radii, conics, opac_row, tw, th)

        def colours_and_packing():
            if sh_rides:
                return  # (already done, in the count pass's scan launch)
            # SH colours (+ depth) and the camera-space normals straight into the packed records (one launch)
            if hm is not None:
                _run(lib.fsgs_sh_fwd_pack_h16, (N, K, sh_degree, ptr(means), ptr(cam["campos"]), ptr(hm["features_dc"]),
                                                ptr(hm["features_rest"]), ptr(radii), ptr(depths), ptr(means2d),
                                                ptr(conics), ptr(opac_sig), ptr(hm["quats"]), ptr(hm["scales"]),
                                                ptr(cam["c2w"]), ptr(packed), ptr(normals_world), ptr(max_last),
                                                max_last.numel(), sp), "fsgs_sh_fwd_split")
                return
            _run(lib.fsgs_sh_fwd_pack, (N, K, sh_degree, ptr(means), ptr(cam["campos"]), ptr(features_dc),
                                        ptr(features_rest), ptr(radii), ptr(depths), ptr(means2d), ptr(conics),
                                        ptr(opac_sig), ptr(quats), ptr(scales), ptr(cam["c2w"]), ptr(packed),
                                        ptr(normals_world), ptr(max_last), max_last.numel(), sp), "fsgs_sh_fwd_split")

        # Work that does not need the lists runs while the host waits for the live total — unless the SH features
        # are still being exchanged between the ranks (info.pre_sh: the trainer's deferred feature update): then
        # the colours come last, so that the exchange overlaps the whole binning + sort chain.
        if pre_sh is None:
            colours_and_packing()
        if direct_bins:
            # (the statistic's word is zeroed by the packing launch: it can be taken only when that ran before the sort)
            gate = long_flag if (pre_sh is None and FWD_WALK.handoff_gate_len > 0) else None
            tpg, isect_ids, flatten_ids, offsets = ops.bin_live_finish(count, means2d, radii, depths, conics, opac_row,
                                                                       tw, th, gate, FWD_WALK.handoff_gate_len)
        else:
            tpg, isect_ids, flatten_ids, offsets = ops.isect_finish_live(count, means2d, radii, depths, conics,
                                                                         opac_row, tw, th, want_ids=False, route_hint=True)
        if pre_sh is not None:
            pre_sh()
            colours_and_packing()
        if info.bin_hist_want and direct_bins and BIN_FILL and N < BIN_FILL_MAX_N and count.get("direct"):
            # the room every tile's bucket gets in this view's next frame
            if count.get("fill"):
                info.bin_route = "fill"
                info.bin_hist_out = {"base": count["next_base"]}
            elif count["offsets"].numel() == tw * th + 1 and hm is None and info.zcut_in is None:
                info.bin_hist_out = {"base": ops.bin_fill_history_from_offsets(count["offsets"], tw * th)}
        rule_diff = 0
        M = flatten_ids.numel()
        no_wait = direct_bins and count.get("capacity", 0) > 0  # M is then the capacity, offsets has T + 1 entries
        if no_wait:
            info.pending_count = count
        else:
            info.n_live = M

        needs_bwd = any(ctx.needs_input_grad[:6])
        a = lambda n: (n + 255) // 256 * 256  # noqa: E731
        n_tiles = tw * th
        if needs_bwd:
            cap, slots = ops.quad_stream_sizes(1, tw, th, M)
            rec_bytes = 4 * cap * 64
            seg_bytes = 4 * slots * 64 * (1 + 4 + 3) * 4
            nrec_bytes = 4 * n_tiles * 4
            order_bytes = BWD_ORDER_CLASSES * nrec_bytes if (bwd_queue is not None and BWD_ORDER) else 0
            arena = WORKSPACE.take(a(rec_bytes) + a(seg_bytes) + 2 * a(nrec_bytes) + order_bytes, dev)
            records = arena[:rec_bytes].view(torch.float32)
            seg_state = arena[a(rec_bytes):a(rec_bytes) + seg_bytes].view(torch.float32)
            o = a(rec_bytes) + a(seg_bytes)
            n_rec = arena[o:o + nrec_bytes].view(torch.int32)
            seg_split = arena[o + a(nrec_bytes):o + a(nrec_bytes) + nrec_bytes].view(torch.int32)
            o += 2 * a(nrec_bytes)
            bwd_order = arena[o:o + order_bytes].view(torch.int32) if order_bytes else None
        else:
            arena = records = seg_state = n_rec = seg_split = bwd_order = None
            bwd_queue = None
        render = torch.empty(1, H, W, 4, **f32)
        alphas = torch.empty(1, H, W, 1, **f32)
        last_ids = torch.empty(1, H, W, dtype=torch.int32, device=dev)
        render_extra = torch.empty(1, H, W, 3, **f32)
        if not direct_bins:
            # (the list chain + radix sort of very dense frames knows no cuts: none applied, none recorded — the caller
            # sees zcut_out = None)
            info.zcut_in = info.zcut_out = None
        track_cuts = info.zcut_out is not None
        tile_open, bad, hit = (ops.zcut_scratch(dev, n_tiles) if (track_cuts or info.zcut_in is not None)
                               else (None, None, None))
        walk, handoff = FWD_WALK.choice(N)
        if not direct_bins or (FWD_WALK.handoff_gate_len > 0 and pre_sh is not None and not FWD_WALK.forced):
            handoff = 0  # (no frame statistic on these routes: the plain one-wave walk)
        info.fwd_walk = walk
        tail, tail_items, epoch = FWD_WALK.scratch(dev) if handoff > 0 else (None, 0, 0)
        _run(lib.fsgs_raster_fwd_quad, (1, 4, ptr(packed), ptr(flatten_ids), ptr(offsets), -M if no_wait else M,
                                       None, W, H, tw, th, 1,
                                       ptr(render), ptr(alphas), ptr(last_ids), ptr(records), ptr(n_rec),
                                       ptr(seg_state), ptr(render_extra), ptr(max_last), ptr(tile_open), walk,
                                       ptr(tail), 0 if tail is None else tail.numel(), int(tail_items), int(handoff),
                                       int(FWD_WALK.handoff_rel_len),
                                       ptr(long_flag) if (direct_bins and pre_sh is None and FWD_WALK.handoff_gate_len > 0)
                                       else None, int(epoch), ptr(seg_split), ptr(bwd_queue),
                                       n_q_items if bwd_queue is not None else 0,
                                       ptr(count.get("tile_order")) if (direct_bins and FWD_TILE_ORDER and walk == 1) else None,
                                       sp),
             "fsgs_raster_fwd_quad", "_d4e3")
        if tile_open is not None:
            # this frame's saturation depths become the next frame's cuts; a cut tile left open spoils the frame
            verdict = ops.zcut_verdict_buffer(dev)
            zout = info.zcut_out if info.zcut_out is not None else torch.empty(n_tiles, **f32)
            _run(lib.fsgs_tile_zcut_update, (tw, th, W, H, ptr(last_ids), ptr(flatten_ids), ptr(depths), ptr(offsets),
                                            ptr(tile_open), ptr(info.zcut_in),
                                            ptr(hit) if info.zcut_in is not None else None, ptr(zout), ptr(bad),
                                            verdict.data_ptr(), float(info.zcut_margins[0]),
                                            float(info.zcut_margins[1]), float(info.zcut_margins[2]), sp),
                 "fsgs_tile_zcut_update")
            info.pending_verdict = verdict if info.zcut_in is not None else None
        rgb = torch.empty(H, W, 3, **f32)
        depth = torch.empty(H, W, 1, **f32)
        normal = torch.empty(H, W, 3, **f32)
        lt = info.loss_targets
        # (riding in the epilogue launch: the backward's dispatch order — the streams' lengths are known now)
        # (classes of 16 records of a quadrant's own share — at most 8 segments where long streams are shared out —, of 128
        # in dense scenes, whose quadrants keep whole streams of hundreds to thousands of records)
        order_args = ((ptr(n_rec), ptr(seg_split), bwd_queue[4 + 4 * n_q_items:].data_ptr(), ptr(bwd_order),
                       4 if n_q_items > 0 else 7, tw, th)
                      if bwd_order is not None else (None, None, None, None, 0, 0, 0))
        if lt is not None:
            # the trainer's tape-free step: the depth / normal L1 terms of its loss are evaluated on the pixels as
            # they are formed (partial sums + gradient images), no separate pass over the two images
            P = H * W
            partial = torch.empty((P + 255) // 256, 2, **f32)
            v_depth_img = torch.empty(H, W, 1, **f32)
            v_normal_img = torch.empty(H, W, 3, **f32) if lt.get("normal") is not None else None
            _run(lib.fsgs_epilogue_fwd_order, (P, ptr(render), ptr(alphas), ptr(render_extra), ptr(background),
                                              ptr(max_last), n_cells, ptr(rgb), ptr(depth), ptr(normal),
                                              ptr(lt["depth"]), ptr(lt.get("normal")), ptr(lt["seed"]),
                                              float(lt["g_depth"]), float(lt["g_normal"]), ptr(partial),
                                              ptr(v_depth_img), ptr(v_normal_img)) + order_args + (sp,),
                 "fsgs_epilogue_fwd")
            info.aux_loss = (partial, v_depth_img, v_normal_img)
        else:
            _run(lib.fsgs_epilogue_fwd_order, (H * W, ptr(render), ptr(alphas), ptr(render_extra), ptr(background),
                                              ptr(max_last), n_cells, ptr(rgb), ptr(depth), ptr(normal), None, None,
                                              None, 0.0, 0.0, None, None, None) + order_args + (sp,), "fsgs_epilogue_fwd")

        info.radii, info.means2d, info.depths, info.conics = radii, means2d, depths, conics
        info.opac_row, info.tiles = opac_row, (tw, th)
        info.tiles_per_gauss, info.isect_ids, info.payload = tpg, isect_ids, flatten_ids
        info.isect_offsets = offsets[:tw * th].view(1, th, tw) if no_wait else offsets
        info.last_ids, info.normals_world = last_ids, normals_world
        info.tile_order = count.get("tile_order") if direct_bins else None  # (tiles longest list first, or None: tests)
        info.legacy_rule_diff = rule_diff
        info.streams = (records, n_rec, seg_state)  # (what the backward reads; valid until it has run: tools / tests)

        if needs_bwd:
            ctx.save_for_backward(means, scales, quats, features_dc, features_rest, scales_exp, opac_sig, radii,
                                  conics, records, n_rec, offsets, render, alphas, last_ids, render_extra, seg_state,
                                  background, seg_split, bwd_queue if bwd_queue is not None else seg_split)
            ctx.has_bwd_queue = bwd_queue is not None
            ctx.bwd_order = bwd_order  # (a view of the frame's arena, like the streams)
            ctx.n_q_items = max(n_q_items, 0)
            ctx.arena = arena
        else:
            WORKSPACE.give(arena)
        ctx.cam = cam
        ctx.dims = (N, K, W, H, tw, th, M, sh_degree)
        ctx.info = info
        ctx.grad_out = grad_out
        return rgb, depth, normal, alphas[0]

    @staticmethod
    def backward(ctx, v_rgb, v_depth, v_normal, v_alpha_out):
        (means, scales, quats, features_dc, features_rest, scales_exp, opac_sig, radii, conics, records, n_rec,
         offsets, render, alphas, last_ids, render_extra, seg_state, background, seg_split, bwd_queue) = ctx.saved_tensors
        if not ctx.has_bwd_queue:
            bwd_queue = None
        n_queue = ctx.n_q_items if bwd_queue is not None else 0
        bwd_order = getattr(ctx, "bwd_order", None) if bwd_queue is not None else None
        N, K, W, H, tw, th, M, sh_degree = ctx.dims
        cam = ctx.cam
        lib = load()
        dev = means.device
        sp = stream_ptr(dev)
        f32 = dict(dtype=torch.float32, device=dev)
        c = lambda t: None if t is None else t.contiguous()  # noqa: E731
        v_rgb, v_depth, v_normal, v_alpha_out = c(v_rgb), c(v_depth), c(v_normal), c(v_alpha_out)

        v_packed = _grad_accumulator(dev, N)  # all zeros: the per-Gaussian backward clears it again after reading
        rep_rows = N if _grad_lines(N) > 1 else 0
        if v_rgb is not None and IMAGE_GRADS_IN_BWD:
            # the image gradients go straight into the compositing backward, which derives v_render / v_alphas /
            # v_render_extra per pixel itself (no epilogue launch, no 32 B/pixel round trip)
            _run(lib.fsgs_raster_bwd_quad_images, (ptr(records), ptr(n_rec), ptr(offsets), M, W, H, tw, th, ptr(render),
                                                   ptr(alphas), ptr(last_ids), ptr(render_extra), ptr(background),
                                                   ptr(v_rgb), ptr(v_depth), ptr(v_normal), ptr(v_alpha_out),
                                                   ptr(seg_state), 1, ptr(v_packed), rep_rows, BWD_DISPATCH.stride(W, H),
                                                   ptr(seg_split), ptr(bwd_queue), n_queue, ptr(bwd_order), sp),
                 "fsgs_raster_bwd_quad", "_d4e3")
        else:
            v_render = torch.empty(1, H, W, 4, **f32)
            v_alphas = torch.empty(1, H, W, 1, **f32)
            v_render_extra = torch.empty(1, H, W, 3, **f32)
            _run(lib.fsgs_epilogue_bwd, (H * W, ptr(render), ptr(alphas), ptr(render_extra), ptr(background), ptr(v_rgb),
                                        ptr(v_depth), ptr(v_normal), ptr(v_alpha_out), ptr(v_render), ptr(v_alphas),
                                        ptr(v_render_extra), sp), "fsgs_epilogue_bwd")
            _run(lib.fsgs_raster_bwd_quad, (1, 4, ptr(records), ptr(n_rec), ptr(offsets), M, None, W, H, tw, th, 1,
                                           ptr(render), ptr(alphas), ptr(last_ids), ptr(v_render), ptr(v_alphas),
                                           ptr(seg_state), 1, ptr(render_extra), ptr(v_render_extra), ptr(v_packed),
                                           0, ptr(seg_split), ptr(bwd_queue), n_queue, ptr(bwd_order), sp),
                 "fsgs_raster_bwd_quad", "_d4e3")
        WORKSPACE.give(getattr(ctx, "arena", None))
        ctx.arena = None

        # final gradients go straight into the caller's buffers (views of the gradient slab) if given
        go = ctx.grad_out or {}

        def out(name, like):
            t = go.get(name)
            return t if t is not None else torch.empty_like(like)

        g_means, g_scales, g_quats = out("means", means), out("scales", scales), out("quats", quats)
        g_dc, g_rest = out("features_dc", features_dc), out("features_rest", features_rest)
        g_opac = out("opacities", opac_sig.view(N, 1))
        factors = ctx.info.sh_factors_out
        hm = ctx.info.half
        v_abs = torch.empty(1, N, 2, **f32)
        stats = ctx.info.stats_out
        if stats is not None and stats["xys_grad_norm"].shape[0] != N:
            stats = None
        stat_args = (ptr(stats["xys_grad_norm"]) if stats else None, ptr(stats["vis_counts"]) if stats else None,
                     ptr(stats["max_2Dsize"]) if stats else None, float(stats["inv_max_hw"]) if stats else 0.0,
                     ptr(ctx.info.frozen), rep_rows, sp)
        if factors is not None:
            # data-parallel step: only the FACTORS of the coefficient gradients are produced here ([N + 1,4]: masked
            # colour gradients + the camera centre); the trainer all-gathers them and fsgs_sh_coeff_grad rebuilds the
            # mean of the ranks' coefficient gradients in the slab (16 B instead of 192 B per Gaussian on the links)
            assert hm is None, "the factored feature exchange reads fp32 features (trainer switches it off)"
            assert tuple(factors.shape) == (N + 1, 4) and factors.is_contiguous()
        # per-launch switches of the per-Gaussian backward (fsgs.h: FSGS_GSB_*): lazy coefficient fetch and idle-element
        # skipping both turn independent loads into dependent ones — they pay where most gradient lines are empty
        gsb_flags = (1 if N >= LAZY_SH_MIN_N else 0) | (2 if N >= SKIP_IDLE_MIN_N else 0)
        aib = ctx.info.adam_in_backward
        if aib is not None and factors is None and K == 16 and sh_degree <= 3 and ONE_LAUNCH_GAUSSIAN_BWD:
            # the step's Adam update rides in this launch: the gradients are applied where they are formed and never
            # written (the caller's gradient buffers are NOT filled; trainer: FSGS_ADAM_IN_BACKWARD)
            import ctypes as C
            f = lib.fsgs_gauss_sh_bwd_adam_h16 if hm is not None else lib.fsgs_gauss_sh_bwd_adam
            src = hm if hm is not None else dict(features_dc=features_dc, features_rest=features_rest, quats=quats,
                                                 scales=scales)
            _run(f, (N, sh_degree, ptr(means), ptr(cam["campos"]), ptr(src["features_dc"]), ptr(src["features_rest"]),
                     ptr(src["quats"]), ptr(src["scales"]), ptr(scales_exp), ptr(opac_sig), ptr(cam["viewmat"]),
                     ptr(cam["K"]), ptr(cam["c2w"]), W, H, 0.3, ptr(radii), ptr(conics), ptr(v_packed), ptr(v_abs))
                 + stat_args[:-1] + (C.byref(aib), float(ctx.info.min_scale_g), gsb_flags, sp), "fsgs_gaussian_bwd")
            ctx.info.adam_applied = True
        elif hm is not None and K == 16 and sh_degree <= 3 and ONE_LAUNCH_GAUSSIAN_BWD:
            _run(lib.fsgs_gauss_sh_bwd_h16,
                 (N, sh_degree, ptr(means), ptr(cam["campos"]), ptr(hm["features_dc"]), ptr(hm["features_rest"]),
                  ptr(hm["quats"]), ptr(hm["scales"]), ptr(scales_exp), ptr(opac_sig), ptr(cam["viewmat"]), ptr(cam["K"]),
                  ptr(cam["c2w"]), W, H, 0.3, ptr(radii), ptr(conics), ptr(v_packed), ptr(g_dc), ptr(g_rest),
                  ptr(g_means), ptr(g_quats), ptr(g_scales), ptr(g_opac), ptr(v_abs)) + stat_args[:-1] + (gsb_flags, sp),
                 "fsgs_gaussian_bwd")
        elif hm is None and K == 16 and sh_degree <= 3 and ONE_LAUNCH_GAUSSIAN_BWD:
            # the SH backward and everything else per Gaussian (projection / normal / activation VJPs, absgrad,
            # statistics) in one launch, one thread per Gaussian
            _run(lib.fsgs_gauss_sh_bwd,
                 (N, sh_degree, ptr(means), ptr(cam["campos"]), ptr(features_dc), ptr(features_rest), ptr(quats),
                  ptr(scales), ptr(scales_exp), ptr(opac_sig), ptr(cam["viewmat"]), ptr(cam["K"]), ptr(cam["c2w"]), W, H,
                  0.3, ptr(radii), ptr(conics), ptr(v_packed), None if factors is not None else ptr(g_dc),
                  None if factors is not None else ptr(g_rest), ptr(factors) if factors is not None else None,
                  ptr(g_means), ptr(g_quats), ptr(g_scales), ptr(g_opac), ptr(v_abs)) + stat_args[:-1] + (gsb_flags, sp),
                 "fsgs_gaussian_bwd")
        else:
            # SH: colour gradients are the first floats of the packed records; writes the view-direction share of v_means
            if hm is not None:
                _run(lib.fsgs_sh_bwd_split_h16, (N, K, sh_degree, ptr(means), ptr(cam["campos"]), ptr(hm["features_dc"]),
                                                 ptr(hm["features_rest"]), ptr(radii), 16, ptr(v_packed), ptr(g_dc),
                                                 ptr(g_rest), ptr(g_means), None, 1, rep_rows, sp),
                     "fsgs_sh_bwd_split")
            elif factors is not None:
                _run(lib.fsgs_sh_bwd_colors, (N, K, sh_degree, ptr(means), ptr(cam["campos"]), ptr(features_dc),
                                              ptr(features_rest), ptr(radii), 16, ptr(v_packed), ptr(factors),
                                              ptr(g_means), 1, rep_rows, sp), "fsgs_sh_bwd_split")
            else:
                _run(lib.fsgs_sh_bwd_split, (1, N, K, sh_degree, ptr(means), ptr(cam["campos"]), ptr(features_dc),
                                             ptr(features_rest), ptr(radii), 16, ptr(v_packed), ptr(g_dc), ptr(g_rest),
                                             ptr(g_means), None, 1, rep_rows, sp), "fsgs_sh_bwd_split")
            # everything else per Gaussian in one launch (projection / normal / activation VJPs, absgrad, statistics)
            _run(lib.fsgs_gaussian_bwd_h16 if hm is not None else lib.fsgs_gaussian_bwd,
                 (N, ptr(means), ptr(hm["quats"] if hm is not None else quats),
                  ptr(hm["scales"] if hm is not None else scales), ptr(scales_exp), ptr(opac_sig), ptr(cam["viewmat"]),
                  ptr(cam["K"]), ptr(cam["c2w"]), W, H, 0.3, ptr(radii), ptr(conics), ptr(v_packed), 1, ptr(g_means),
                  ptr(g_quats), ptr(g_scales), ptr(g_opac), ptr(v_abs)) + stat_args, "fsgs_gaussian_bwd")
        ctx.info.absgrad = v_abs
        ctx.info.stats_done = stats is not None
        if ctx.grad_out:
            # already written into the caller's gradient buffers (the trainer's slab views, which ARE the
            # parameters' .grad): returning them as well would make autograd add them a second time
            return (None,) * 13
        return (g_means, g_scales, g_quats, g_dc, g_rest, g_opac, None, None, None, None, None, None, None)


def render_fusionsense_fused(gauss_params: Dict[str, Tensor], camera: Camera, sh_degree: int = 3,
                             background: Optional[Tensor] = None, device: Optional[torch.device] = None,
                             grad_out: Optional[Dict[str, Tensor]] = None,
                             stats_out: Optional[dict] = None, add_mask: Optional[Tensor] = None,
                             crop_box=None, training: bool = True,
                             binary_threshold: Optional[float] = None,
                             half: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
    """Same outputs as :func:`fusionsense_amd.fusion.render_fusionsense` (rgb, depth, normal,
    accumulation, radii, normals_world, ...), computed by one fused autograd node.  ``add_mask``,
    ``crop_box`` / ``training`` and ``binary_threshold`` have the meaning they have there
    (dn_model.py:535-541, :505-532, :492-503)."""
    dev = device or gauss_params["means"].device
    if background is None:
        background = _ONES3.get(str(dev))
        if background is None:
            background = _ONES3.setdefault(str(dev), torch.ones(3, device=dev))
    if crop_box is not None and not training:
        from .crop import crop_params, get_empty_outputs
        if binary_threshold is not None:  # the write precedes the crop in get_outputs
            from .splatfacto import binary_opacity_write_
            binary_opacity_write_(gauss_params["opacities"], binary_threshold)
            binary_threshold = None
        crop_ids = crop_box.within(gauss_params["means"].detach())
        if crop_ids.sum() == 0:
            return get_empty_outputs(camera.width, camera.height, background)
        gauss_params = crop_params(gauss_params, crop_ids)
        grad_out = None
        if half is not None:  # (the binary write above went to the masters: re-derive the cropped mirrors from them)
            half = {k: gauss_params[k].detach().to(torch.float16).contiguous() for k in half}
    cam = _camera_on_device(camera, dev)
    info = FrameInfo()
    info.stats_out = stats_out
    info.binary_threshold = binary_threshold
    info.half = half
    if add_mask is not None:
        info.frozen = add_mask.to(device=dev, dtype=torch.uint8).contiguous()
    rgb, depth, normal, alpha = _FusedGetOutputs.apply(
        gauss_params["means"], gauss_params["scales"], gauss_params["quats"], gauss_params["features_dc"],
        gauss_params["features_rest"], gauss_params["opacities"], cam, camera.width, camera.height, int(sh_degree),
        background.contiguous(), info, grad_out)
    return {"rgb": rgb, "depth": depth, "normal": normal, "accumulation": alpha, "background": background,
            "info": info, "xys": info, "radii": info.radii[0], "normals_world": info.normals_world}


class _DirectCtx:
    """What ``_FusedGetOutputs`` / ``ops._TrainLoss`` need from an autograd context, without the tape: the
    trainer's step is a fixed two-node graph, so it calls forward and backward of both nodes itself and
    skips torch.autograd's bookkeeping and the hand-off to its engine thread (~15 us of GPU bubble in front
    of the first backward launch, ~80 us of host time per step)."""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = tuple(needs_input_grad)
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def set_materialize_grads(self, value):
        pass

    def mark_non_differentiable(self, *args):
        pass


def fused_step_forward_backward(gauss_params: Dict[str, Tensor], camera: Camera, target: Dict[str, Tensor],
                                sh_degree: int, device, grad_out: Dict[str, Tensor], seed_grad: Tensor,
                                stats_out: Optional[dict] = None, add_mask: Optional[Tensor] = None,
                                binary_threshold: Optional[float] = None, ssim_lambda: float = 0.2,
                                w_depth: float = 0.2, w_normal: float = 0.1, pre_sh=None, adam_rider=None, sh_factors_out=None,
                                fusion=None, half: Optional[Dict[str, Tensor]] = None, live_capacity: int = 0,
                                zcut_in: Optional[Tensor] = None, zcut_out: Optional[Tensor] = None,
                                zcut_margins=None, adam_in_backward=None, bin_hist: Optional[dict] = None,
                                want_bin_hist: bool = False):
    """get_outputs -> loss -> both backward passes, without the autograd tape.  The parameter gradients land in
    ``grad_out`` (the trainer's slab views).  Returns (loss 0-d tensor, outputs dict).
    ``target`` is either the benchmark targets of BASELINE config #2 (dict rgb / depth / normal: L1 + SSIM on rgb,
    plain L1 on depth and normal) or, with ``fusion = (LossConfig, touch_idx, touch_normals)``, a
    prepared FrameBatch: the reference's get_loss_dict (dn_model.py:673-925, ops._FusionLoss)."""
    dev = device
    background = _ONES3.get(str(dev))
    if background is None:
        background = _ONES3.setdefault(str(dev), torch.ones(3, device=dev))
    cam = _camera_on_device(camera, dev)
    info = FrameInfo()
    info.stats_out = stats_out
    info.binary_threshold = binary_threshold
    info.pre_sh = pre_sh
    info.adam_rider = adam_rider
    info.sh_factors_out = sh_factors_out
    info.half = half
    info.live_capacity = int(live_capacity)
    info.zcut_in, info.zcut_out = zcut_in, zcut_out
    info.adam_in_backward = adam_in_backward
    info.bin_hist_in, info.bin_hist_want = bin_hist, bool(want_bin_hist)
    if zcut_margins is not None:
        info.zcut_margins = zcut_margins

    def check_live_total(ctx):
        """No-wait binning: the forward's launches are enqueued, now look at the frame's live total (it has normally
        long arrived).  On overflow the truncated frame is abandoned before any BACKWARD launch (statistics, gradient
        accumulator, gradient slab, the step's own Adam).  What the count pass has already done by then is final and
        must not be done twice: the binary-opacity write to the logits (master and half mirror), a riding feature
        Adam step of the PREVIOUS frame and a flushed deferred update (``pre_sh``) — the trainer's retry therefore
        runs without ``binary_threshold`` (the write is not idempotent for thresholds outside (0, 1]) and finds no
        pending update."""
        if info.pending_count is not None:
            try:
                info.n_live = ops.bin_live_check(info.pending_count)
            except ops.LiveListOverflow:
                WORKSPACE.give(getattr(ctx, "arena", None))
                ctx.arena = None
                raise
            info.pending_count = None
            info.payload = info.payload[:info.n_live]
        if info.pending_verdict is not None:
            # occlusion cuts: the verdict lands right after the forward walk; a cut tile that did not saturate inside
            # its prefix -> the images miss contributions, the frame is abandoned like an overflowed one
            ok = ops.zcut_check(info.pending_verdict)
            info.pending_verdict = None
            if ok == 2:  # open tiles under cuts: exact only if none of them lost a live pair
                ok = 0 if ops.zcut_recheck(info.means2d, info.radii, info.depths, info.conics, info.opac_row,
                                           info.tiles[0], info.tiles[1], info.zcut_in) else 1
            ok = ok == 0
            if not ok:
                WORKSPACE.give(getattr(ctx, "arena", None))
                ctx.arena = None
                raise ops.OcclusionCutInvalid("a cut tile did not saturate inside its depth prefix")

    if add_mask is not None:
        info.frozen = add_mask.to(device=dev, dtype=torch.uint8).contiguous()
    if fusion is not None:
        with torch.no_grad():
            cfg, touch_idx, touch_normals = fusion
            ctx = _DirectCtx((True,) * 6 + (False,) * 7)
            rgb, depth, normal, alpha = _FusedGetOutputs.forward(
                ctx, gauss_params["means"], gauss_params["scales"], gauss_params["quats"], gauss_params["features_dc"],
                gauss_params["features_rest"], gauss_params["opacities"], cam, camera.width, camera.height,
                int(sh_degree), background, info, grad_out)
            lctx = _DirectCtx((True, True, True, False) + (False,) * 7)
            loss = ops._FusionLoss.forward(lctx, rgb, depth, normal, gauss_params["scales"].data, target, cfg,
                                           info.normals_world, touch_idx, touch_normals, seed_grad, True,
                                           (camera.fx, camera.fy, camera.cx, camera.cy))
            check_live_total(ctx)
            g_min = lctx.g[3]
            info.min_scale_g = g_min  # (an Adam step carried by the backward takes the term's gradient with it)
            v = ops._FusionLoss.backward(lctx, seed_grad)
            _FusedGetOutputs.backward(ctx, v[0], v[1], v[2], None)
            if g_min != 0.0 and not info.adam_applied:
                # the min-scale term reaches the log-scales directly, touch anchors included (:817-819)
                ops.min_scale_grad_(gauss_params["scales"].data, g_min, seed_grad.reshape(1), grad_out["scales"])
        out = {"rgb": rgb, "depth": depth, "normal": normal, "accumulation": alpha, "background": background,
               "info": info, "xys": info, "radii": info.radii[0], "normals_world": info.normals_world}
        return loss, out
    has_n = "normal" in target
    n_pix = camera.width * camera.height
    info.loss_targets = dict(depth=target["depth"].contiguous(), normal=target["normal"].contiguous() if has_n else None,
                             seed=seed_grad, g_depth=float(w_depth) / n_pix,
                             g_normal=float(w_normal) / (3.0 * n_pix) if has_n else 0.0)
    with torch.no_grad():
        ctx = _DirectCtx((True,) * 6 + (False,) * 7)
        rgb, depth, normal, alpha = _FusedGetOutputs.forward(
            ctx, gauss_params["means"], gauss_params["scales"], gauss_params["quats"], gauss_params["features_dc"],
            gauss_params["features_rest"], gauss_params["opacities"], cam, camera.width, camera.height,
            int(sh_degree), background, info, grad_out)
        lctx = _DirectCtx((True, False, True, False, has_n, False, False, False, False))
        loss = ops._TrainLoss.forward(lctx, rgb, target["rgb"], depth, target["depth"], normal if has_n else None,
                                      target["normal"] if has_n else None, float(ssim_lambda), float(w_depth),
                                      float(w_normal), seed=seed_grad, aux_done=info.aux_loss)
        check_live_total(ctx)
        v = ops._TrainLoss.backward(lctx, seed_grad)
        _FusedGetOutputs.backward(ctx, v[0], v[2], v[4], None)
    out = {"rgb": rgb, "depth": depth, "normal": normal, "accumulation": alpha, "background": background,
           "info": info, "xys": info, "radii": info.radii[0], "normals_world": info.normals_world}
    return loss, out
