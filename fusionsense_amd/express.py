"""The trainer's step as ONE library call (include/fsgs.h: fsgs_step_plan; csrc/step.hip).

``fused.fused_step_forward_backward`` drives a step launch by launch from Python: ~25 tensor allocations, ~9 ctypes
calls with 20-45 converted arguments each, the autograd-free contexts — 0.43-0.48 ms of host time for a step the GPU
runs in 0.49 ms (config #2), so the host is never ahead and every hiccup of it is a GPU bubble.  Here the same launches,
with the same arguments, are enqueued by ``fsgs_step_run`` from one argument block that lives across steps:

* every internal buffer of a frame is allocated once per (model size, frame shape, list capacity) and reused; what a
  caller may keep — the loss scalar and the rgb / depth / normal / accumulation images — are fresh tensors every step, as
  on the per-op route; the frame's side outputs (``info``: projection results, lists, ``absgrad`` ...) exist twice and
  alternate, so they stay valid until the next-but-one step (comparing two consecutive steps works as it does there);
* per step only what changes is written into the block: the camera's and targets' pointers, the view's bucket room,
  the Adam step number and the means' learning rate, the statistics' pointers;
* the one host wait of the step (the live total / bucket-overflow verdict, in mapped memory) happens inside the call,
  between the loss forward and the first launch with side effects, exactly where ``check_live_total`` has it.

It applies to the frames the count-free binning route applies to (a revisited view, fp32 storage, fewer than 2^20
Gaussians, no occlusion cuts) with the benchmark loss (BASELINE config #2) on one rank with the Adam step
inside the backward; every other frame — and the redo of an overflowed one — takes ``fused_step_forward_backward``.
Mirrors /root/reference/dn_splatter/dn_model.py:469-671 (get_outputs) + nerfstudio's after_train (registered at
:1385-1389) + the Adam groups of dn_config.py:36-75, as the per-op route does."""
from __future__ import annotations

import ctypes as C
import math
import time
from typing import Dict, Optional

import torch
from torch import Tensor

from . import fused, ops
from ._lib import StepPlan, check, load, stream_ptr

SPIN_LIMIT = 50_000_000  # polls of the mapped flag before the call falls back to a stream synchronise (~ seconds)
# launch names as ops.TIMER / bench.py know them (the per-op route's span names), in the order of the step
LAUNCHES = ("isect_count_live", "tile_sort", "raster_fwd_quad_d4e3", "epilogue_fwd", "ssim_l1_fwd", "ssim_l1_bwd",
            "raster_bwd_quad_d4e3", "gaussian_bwd", "fusion_aux_loss")


def fused_param_order():
    from .trainer import PARAM_ORDER
    return PARAM_ORDER


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


class ExpressStep:
    """Buffers + two argument blocks for one (N, W, H, capacity, normal target?) shape of a trainer's frames."""

    def __init__(self, trainer, W: int, H: int, capacity: int, has_normal: bool):
        lib = load()
        dev = trainer.device
        N = trainer.num_gaussians()
        self.key = (N, W, H, int(capacity), bool(has_normal))
        # module switches the buffers' layout depends on (tests and A/B runs move them): a change re-makes the blocks
        self.switches = self._switches()
        self.N, self.W, self.H, self.capacity = N, W, H, int(capacity)
        tw, th = math.ceil(W / fused.TILE), math.ceil(H / fused.TILE)
        self.tw, self.th, self.T = tw, th, tw * th
        T, P = self.T, W * H
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        assert int(lib.fsgs_step_plan_bytes()) == C.sizeof(StepPlan), "fsgs_step_plan: binding and library disagree"
        # ---- scratch of a frame (one copy) -------------------------------------------------------------------------------
        self.packed = torch.empty(N, 16, **f32)
        self.words = ops.bin_fill_words(self.capacity, T)
        self.buckets = torch.empty(8 * self.words, dtype=torch.uint8, device=dev)
        cursor = ops._BIN_CURSORS.get((str(dev), T))
        if cursor is None:
            cursor = ops._BIN_CURSORS[(str(dev), T)] = torch.zeros(T, **i32)
        self.cursor = cursor
        self.tile_order = torch.empty(T, **i32)
        self.n_cells = int(lib.fsgs_raster_quad_max_cells())
        self.n_q_items = fused.BWD_QUEUE_ITEMS if N < fused.FWD_WALK.handoff_max_n else (0 if fused.BWD_ORDER_DENSE else -1)
        q_words = 4 + 4 * max(self.n_q_items, 0) + fused.BWD_ORDER_CLASSES
        self.max_last = torch.empty(self.n_cells + q_words + 4, **f32)
        self.bwd_queue = (self.max_last[self.n_cells:self.n_cells + q_words].view(torch.int32)
                          if (fused.BWD_QUEUE_ITEMS > 0 and self.n_q_items >= 0) else None)
        self.long_flag = self.max_last[self.n_cells + q_words:].view(torch.int32)
        cap, slots = ops.quad_stream_sizes(1, tw, th, self.capacity)
        self.records = torch.empty(4 * cap * 16, **f32)
        self.seg_state = torch.empty(4 * slots * 64 * (1 + 4 + 3), **f32)
        self.n_rec = torch.empty(4 * T, **i32)
        self.seg_split = torch.empty(4 * T, **i32)
        self.bwd_order = (torch.empty(fused.BWD_ORDER_CLASSES * 4 * T, **i32)
                          if (self.bwd_queue is not None and fused.BWD_ORDER) else None)
        self.ssim_maps = torch.empty(3, H, W, 3, **f32)
        self.ssim_rows = int(lib.fsgs_ssim_l1_num_partials(H, W))
        self.ssim_sums = torch.empty(self.ssim_rows, 2, **f32)
        self.aux_rows = (P + 255) // 256
        self.aux_partial = torch.empty(self.aux_rows, 2, **f32)
        self.v_depth_img = torch.empty(H, W, 1, **f32)
        self.v_normal_img = torch.empty(H, W, 3, **f32) if has_normal else None
        # get_loss_dict's terms (loss_kind 1): the depth / normal columns, the min-scale and the touch-normal partial sums
        self.fa_rows = int(lib.fsgs_fusion_aux_num_partials(H, W))
        self.fa_partial = self.ms_partial = self.touch_partial = None  # (made on the first FrameBatch step)
        self.v_rgb = torch.empty(H, W, 3, **f32)
        self.v_packed = fused._grad_accumulator(dev, N)
        self.rep_rows = N if fused._grad_lines(N) > 1 else 0
        self.background = fused._ONES3.get(str(dev))
        if self.background is None:
            self.background = fused._ONES3.setdefault(str(dev), torch.ones(3, device=dev))
        # ---- a frame's outputs, twice ------------------------------------------------------------------------------------
        self.frames = []
        for _ in range(2):
            fr = dict(scales_exp=torch.empty(N, 3, **f32), opac_sig=torch.empty(N, **f32), radii=torch.empty(1, N, **i32),
                      means2d=torch.empty(1, N, 2, **f32), depths=torch.empty(1, N, **f32), conics=torch.empty(1, N, 3, **f32),
                      tpg=torch.empty(1, N, **i32), offsets=torch.empty(T + 1, **i32), normals_world=torch.empty(N, 3, **f32),
                      payload=torch.empty(self.capacity, **i32), render=torch.empty(1, H, W, 4, **f32),
                      last_ids=torch.empty(1, H, W, **i32), render_extra=torch.empty(1, H, W, 3, **f32),
                      absgrad=torch.empty(1, N, 2, **f32),
                      mapped=torch.zeros(4, dtype=torch.int32).pin_memory())
            fr["mapped_np"] = fr["mapped"].numpy()
            fr["plan"] = self._plan(trainer, fr)
            self.frames.append(fr)
        self.turn = 0
        self.last_wait_ms = 0.0
        self.n_live_out = C.c_int64(0)
        # the Adam step of all six groups as the per-Gaussian backward applies it (moments created if this is the first step)
        self.adam = trainer._fused_adam_step(fused_param_order(), getattr(trainer, "adam_steps", 0) + 1, as_groups=True)

    def _plan(self, trainer, fr) -> StepPlan:
        p = StepPlan()
        N, W, H, T = self.N, self.W, self.H, self.T
        p.n, p.sh_degree, p.width, p.height, p.tile_width, p.tile_height = N, 0, W, H, self.tw, self.th
        p.capacity = self.capacity
        pr = trainer._params
        p.means, p.quats, p.log_scales = _p(pr["means"].data), _p(pr["quats"].data), _p(pr["scales"].data)
        p.opac_logit, p.features_dc, p.features_rest = (_p(pr["opacities"].data), _p(pr["features_dc"].data),
                                                         _p(pr["features_rest"].data))
        p.background = _p(self.background)
        p.scales_exp, p.opac_sig, p.radii = _p(fr["scales_exp"]), _p(fr["opac_sig"]), _p(fr["radii"])
        p.means2d, p.depths, p.conics = _p(fr["means2d"]), _p(fr["depths"]), _p(fr["conics"])
        p.tiles_per_gauss, p.isect_offsets = _p(fr["tpg"]), _p(fr["offsets"])
        p.tile_cursor, p.buckets, p.bucket_words = _p(self.cursor), _p(self.buckets), self.words
        p.growth, p.slack, p.mapped = float(ops.BIN_FILL_GROWTH), int(ops.BIN_FILL_SLACK), fr["mapped"].data_ptr()
        p.packed, p.normals_world = _p(self.packed), _p(fr["normals_world"])
        p.zero_cells, p.n_zero, p.max_last = _p(self.max_last), int(self.max_last.numel()), _p(self.max_last)
        p.tile_order = _p(self.tile_order) if ops.BIN_FILL_SORT_ORDER else None
        p.payload = _p(fr["payload"])
        p.render, p.last_ids, p.render_extra = _p(fr["render"]), _p(fr["last_ids"]), _p(fr["render_extra"])
        p.records, p.n_rec, p.seg_state, p.seg_split = _p(self.records), _p(self.n_rec), _p(self.seg_state), _p(self.seg_split)
        p.bwd_queue, p.bwd_queue_items = _p(self.bwd_queue), max(self.n_q_items, 0)
        p.n_cells = self.n_cells  # (rgb / depth / normal / alphas / loss: fresh tensors per step, see run())
        p.aux_partial, p.v_depth_img, p.v_normal_img = _p(self.aux_partial), _p(self.v_depth_img), _p(self.v_normal_img)
        if self.bwd_order is not None:
            p.order_counters = self.bwd_queue[4 + 4 * max(self.n_q_items, 0):].data_ptr()
            p.bwd_order = _p(self.bwd_order)
            p.order_shift = 4 if self.n_q_items > 0 else 7
        p.ssim_maps, p.ssim_sums, p.ssim_rows, p.aux_rows = _p(self.ssim_maps), _p(self.ssim_sums), self.ssim_rows, self.aux_rows
        p.v_rgb = _p(self.v_rgb)
        p.v_packed, p.replica_rows, p.dispatch_stride = _p(self.v_packed), self.rep_rows, fused.BWD_DISPATCH.stride(W, H)
        p.absgrad = _p(fr["absgrad"])
        p.gsb_flags = (1 if N >= fused.LAZY_SH_MIN_N else 0) | (2 if N >= fused.SKIP_IDLE_MIN_N else 0)
        p.armed = 0
        return p

    @staticmethod
    def _switches():
        return (fused.BWD_QUEUE_ITEMS, fused.BWD_ORDER, fused.BWD_ORDER_DENSE, fused.FWD_WALK.handoff_max_n,
                ops.BIN_FILL_SORT_ORDER, ops.BIN_FILL_GROWTH, ops.BIN_FILL_SLACK, fused.GRAD_REPLICAS_MAX_N)

    def still_fits(self, trainer) -> bool:
        """The blocks hold raw pointers: parameters / moments that were re-created (densify, prune, a loaded checkpoint)
        make them stale."""
        pr = trainer._params
        if pr["means"].shape[0] != self.N or self.v_packed is not fused._ACCUM.get(str(trainer.device)):
            return False
        if self.switches != self._switches():
            return False
        a = self.adam
        for k, name in enumerate(fused_param_order()):
            par = pr[name]
            st = trainer.optimizers[name].state.get(par)
            if (st is None or "exp_avg" not in st or a.params[k] != par.data_ptr() or a.exp_avg[k] != st["exp_avg"].data_ptr()
                    or a.exp_avg_sq[k] != st["exp_avg_sq"].data_ptr()):
                return False
        return True

    def run(self, trainer, camera, target: Dict[str, Tensor], hist: dict, stats: Optional[dict], frozen: Optional[Tensor],
            binary_threshold: Optional[float], adam, sh_degree: int, seed: Tensor, ssim_lambda: float = 0.2,
            w_depth: float = 0.2, w_normal: float = 0.1, fusion=None):
        """One step.  Returns (overflowed: bool, n_live, loss 0-d tensor, outputs dict).  On overflow nothing with side
        effects has run except the binary-opacity write of launch 1 (the caller's redo must not threshold again).
        ``fusion`` = (LossConfig, touch_idx, touch_normals): ``target`` is a prepared FrameBatch and the loss is
        get_loss_dict's (the terms ops._FusionLoss evaluates for FusionSense's configuration); None: the benchmark loss."""
        lib = load()
        dev = trainer.device
        fr = self.frames[self.turn]
        self.turn ^= 1
        p = fr["plan"]
        W, H, T = self.W, self.H, self.T
        # what the caller may keep: fresh every step (five small allocations from torch's cache)
        f32 = dict(dtype=torch.float32, device=dev)
        rgb, depth, normal = torch.empty(H, W, 3, **f32), torch.empty(H, W, 1, **f32), torch.empty(H, W, 3, **f32)
        alphas, loss = torch.empty(1, H, W, 1, **f32), torch.empty((), **f32)
        p.rgb, p.depth, p.normal, p.alphas, p.loss_out = (rgb.data_ptr(), depth.data_ptr(), normal.data_ptr(),
                                                          alphas.data_ptr(), loss.data_ptr())
        cam = fused._camera_on_device(camera, dev)
        p.viewmat, p.K, p.campos, p.c2w = _p(cam["viewmat"]), _p(cam["K"]), _p(cam["campos"]), _p(cam["c2w"])
        p.sh_degree = int(sh_degree)
        p.binarise = 0 if binary_threshold is None else 1
        p.binary_threshold = 0.0 if binary_threshold is None else float(binary_threshold)
        # the view's bucket room: base of its previous frame in, room for its next frame out (two buffers per view)
        base = hist["base"]
        nxt = hist.get("spare")
        if nxt is None or nxt.numel() != T + 1:
            nxt = torch.empty(T + 1, dtype=torch.int32, device=dev)
        p.bucket_base, p.next_bucket_base = _p(base), _p(nxt)
        # the forward's walk: hand-off parameters as the per-op route takes them (fused._FusedGetOutputs.forward)
        walk, handoff = fused.FWD_WALK.choice(self.N)
        gate_on = fused.FWD_WALK.handoff_gate_len > 0
        p.long_flag = _p(self.long_flag) if gate_on else None
        p.rel_gate = int(fused.FWD_WALK.handoff_gate_len)
        if handoff > 0:
            tail, items, epoch = fused.FWD_WALK.scratch(dev)
            p.tail_scratch, p.tail_scratch_bytes, p.tail_items, p.tail_epoch = tail.data_ptr(), tail.numel(), int(items), int(epoch)
        else:
            p.tail_scratch, p.tail_scratch_bytes, p.tail_items, p.tail_epoch = None, 0, 0, 0
        p.handoff_records, p.handoff_rel_len = int(handoff), int(fused.FWD_WALK.handoff_rel_len)
        p.seed = _p(seed)
        if fusion is None:
            # targets and loss weights of the benchmark loss (ops._TrainLoss with the epilogue's riding L1 terms)
            n_pix = W * H
            gt_n = target.get("normal")
            p.loss_kind = 0
            p.gt_rgb, p.gt_depth, p.gt_normal = _p(target["rgb"]), _p(target["depth"]), _p(gt_n)
            p.g_depth = float(w_depth) / n_pix
            p.g_normal = float(w_normal) / (3.0 * n_pix) if gt_n is not None else 0.0
            p.g_l1 = (1.0 - ssim_lambda) / (3.0 * H * W)
            p.g_ssim = -ssim_lambda / (3.0 * (H - 10) * (W - 10))
            p.ssim_lambda = float(ssim_lambda)
        else:
            # get_loss_dict (ops._FusionLoss, default terms): ``target`` is the view's FrameBatch
            cfg, touch_idx, touch_normals = fusion
            fb = target
            n_touch = 0 if touch_idx is None else int(touch_idx.numel())
            (g_l1, g_ssim), w_aux, g_min, g_touch = ops.fusion_loss_weights(cfg, fb, self.N, n_touch)
            flags = ops.fusion_aux_flags(cfg)
            if fb.normal is None:
                flags &= ~1
            if self.fa_partial is None:
                self.fa_partial = torch.empty(self.fa_rows, 8, **f32)
                self.ms_partial = torch.empty((self.N + 255) // 256, 2, **f32)
            if n_touch > 0 and (self.touch_partial is None or self.touch_partial.shape[0] != (n_touch + 255) // 256):
                self.touch_partial = torch.empty((n_touch + 255) // 256, 2, **f32)
            p.loss_kind = 1
            p.gt_rgb, p.gt_depth, p.gt_normal = _p(fb.image), None, _p(fb.normal)
            p.mask, p.sensor_depth, p.depth_tol, p.fa_flags = _p(fb.mask), _p(fb.sensor_depth), float(cfg.depth_tolerance), int(flags)
            for k in range(7):
                p.w_aux[k] = float(w_aux[k])
            p.fa_partial, p.fa_rows = _p(self.fa_partial), self.fa_rows
            p.ms_partial = _p(self.ms_partial) if g_min != 0.0 else None
            p.ms_rows, p.g_min = (self.N + 255) // 256, float(g_min)
            p.n_touch = n_touch
            p.touch_idx = _p(touch_idx) if n_touch > 0 else None
            self._touch_normals = touch_normals.contiguous() if n_touch > 0 else None  # (kept alive for the launch)
            p.touch_normals = _p(self._touch_normals)
            p.touch_partial = _p(self.touch_partial) if n_touch > 0 else None
            p.touch_rows, p.g_touch = (n_touch + 255) // 256, float(g_touch)
            p.g_l1, p.g_ssim, p.ssim_lambda = float(g_l1), float(g_ssim), float(cfg.ssim_lambda)
            p.g_depth = p.g_normal = 0.0
        # after_train statistics, touch anchors, Adam
        if stats is not None and stats["xys_grad_norm"].shape[0] == self.N:
            p.xys_grad_norm, p.vis_counts, p.max_2Dsize = (_p(stats["xys_grad_norm"]), _p(stats["vis_counts"]),
                                                           _p(stats["max_2Dsize"]))
            p.inv_max_hw = float(stats["inv_max_hw"])
            stats_on = True
        else:
            p.xys_grad_norm = p.vis_counts = p.max_2Dsize = None
            p.inv_max_hw = 0.0
            stats_on = False
        p.frozen = _p(frozen)
        p.dispatch_stride = fused.BWD_DISPATCH.stride(W, H)
        p.gsb_flags = (1 if self.N >= fused.LAZY_SH_MIN_N else 0) | (2 if self.N >= fused.SKIP_IDLE_MIN_N else 0)
        if adam is not None:
            p.adam = adam
        else:
            # a step WITHOUT its optimizer: the per-Gaussian backward writes the gradients into the trainer's slab views
            p.adam.n_groups = 0
            gv = trainer.slab.views
            p.g_means, p.g_log_scales, p.g_quats = _p(gv["means"]), _p(gv["scales"]), _p(gv["quats"])
            p.g_features_dc, p.g_features_rest, p.g_opac_logit = _p(gv["features_dc"]), _p(gv["features_rest"]), _p(gv["opacities"])
        p.min_scale_g = 0.0
        timed = self._arm_timer(p)
        sp = stream_ptr(dev)
        if timed is None:
            rc = lib.fsgs_step_run(C.byref(p), SPIN_LIMIT, C.byref(self.n_live_out), sp)
        else:
            rc = lib.fsgs_step_run(C.byref(p), SPIN_LIMIT, C.byref(self.n_live_out), sp)
            self._file_timer(p, timed)
        self.last_wait_ms = p.wait_ns * 1e-6  # (host time inside the call spent waiting for the frame's verdict)
        if rc < 0:
            check(rc, "fsgs_step_run")
        n_live = int(self.n_live_out.value)
        info = fused.FrameInfo()
        info.n_live = n_live
        info.bin_route = "fill"
        if rc > 0:  # FSGS_STEP_OVERFLOW: lists truncated / a tile outgrew its bucket — the caller redoes the frame
            return True, n_live, None, {"info": info}
        hist_out = {"base": nxt, "spare": base}
        info.bin_hist_out = hist_out
        info.radii, info.means2d, info.depths, info.conics = fr["radii"], fr["means2d"], fr["depths"], fr["conics"]
        info.opac_row, info.tiles = fr["opac_sig"].view(1, self.N), (self.tw, self.th)
        info.tiles_per_gauss, info.isect_ids = fr["tpg"], None
        info.payload = fr["payload"][:n_live]
        info.isect_offsets = fr["offsets"][:T].view(1, self.th, self.tw)
        info.last_ids, info.normals_world = fr["last_ids"], fr["normals_world"]
        info.tile_order = self.tile_order
        info.streams = (self.records, self.n_rec, self.seg_state)
        info.fwd_walk = walk
        info.absgrad = fr["absgrad"]
        info.stats_out = stats
        info.stats_done = stats_on
        info.adam_applied = adam is not None
        out = {"rgb": rgb, "depth": depth, "normal": normal, "accumulation": alphas[0],
               "background": self.background, "info": info, "xys": info, "radii": fr["radii"][0],
               "normals_world": fr["normals_world"]}
        return False, n_live, loss, out

    # -- bench.py's kernel timer (ops.TIMER): HIP events around the launches it asks for, recorded by the library ----------
    def _arm_timer(self, p: StepPlan):
        t = ops.TIMER
        for k in range(9):
            p.ev_before[k] = None
            p.ev_after[k] = None
        if not t.enabled:
            return None
        timed = []
        for k, name in enumerate(LAUNCHES):
            if k == 8 and p.loss_kind != 1:
                continue
            if t.only is not None and not name.startswith(t.only):
                continue
            if t.every > 1:
                t.seen += 1
                if (t.seen - 1) % t.every != 0:
                    continue
            a, b = t._event(), t._event()
            if not t.pool:  # (not pre-instantiated: a HIP event exists from its first record on)
                a.record()
                b.record()
            p.ev_before[k], p.ev_after[k] = a.cuda_event, b.cuda_event
            timed.append((name, a, b))
        return timed or None

    def _file_timer(self, p: StepPlan, timed) -> None:
        for name, a, b in timed:
            ops.TIMER.records.append((name, a, b, 0.0))
        for k in range(9):
            p.ev_before[k] = None
            p.ev_after[k] = None
