"""Densify / prune host logic: the SplatfactoModel methods FusionSense inherits or overrides
(SURVEY.md §8a-12, 8a-13, Appendix A.1/A.2), operating on a :class:`SplatTrainer`.

Reference behaviour mirrored (same names, argument meaning and order of operations):
  * ``after_train``            nerfstudio SplatfactoModel (registered dn_splatter/dn_model.py:1385-1389)
  * ``refinement_after``       dn_splatter/dn_model.py:326-451
  * ``split_gaussians`` / ``dup_gaussians`` / ``cull_gaussians`` / ``dup_in_optim`` /
    ``remove_from_optim``      nerfstudio SplatfactoModel (called at dn_model.py:369,379,398-401,415,426)

The mask arithmetic is a handful of elementwise torch ops over [N]; everything that moves
rows (62 parameter floats + 2x59 Adam-moment floats per Gaussian) or touches per-Gaussian
statistics goes through libfsgs.so (fsgs_densify_stats, fsgs_split_samples, fsgs_mask_scan,
fsgs_compact_rows): one order-preserving compaction per tensor straight into the grown
allocation instead of the reference's ``torch.cat`` followed by ``param[~culls]``.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import Tensor

from . import ops


@dataclass
class SplatfactoConfig:
    """nerfstudio 1.1.3 SplatfactoModelConfig defaults with the overrides of
    DNSplatterModelConfig (dn_model.py:116-141) and configs/config.py:3-7."""
    warmup_length: int = 500
    refine_every: int = 100
    cull_alpha_thresh: float = 0.1
    cull_scale_thresh: float = 0.5
    continue_cull_post_densification: bool = True
    reset_alpha_every: int = 30
    densify_grad_thresh: float = 0.0008
    densify_size_thresh: float = 0.01
    n_split_samples: int = 2
    sh_degree_interval: int = 1000
    cull_screen_size: float = 0.15
    split_screen_size: float = 0.05
    stop_screen_size_at: int = 4000
    stop_split_at: int = 10000
    sh_degree: int = 3
    use_binary_opacities: bool = True        # dn_model.py:94
    binary_opacities_threshold: float = 0.9  # dn_model.py:96 (compared with the RAW opacity parameter)


def binary_opacity_active(cfg: SplatfactoConfig, step: int) -> bool:
    """Whether get_outputs overwrites ``opacities.data`` with {0, 1} at this step (dn_model.py:492-503):
    after the warm-up, except on the opacity-reset steps and the 200 steps that follow each of them."""
    if not (cfg.use_binary_opacities and step > cfg.warmup_length):
        return False
    skip_steps = cfg.reset_alpha_every * cfg.refine_every
    margin = 200
    return (step % skip_steps != 0) and (step % skip_steps not in range(1, margin + 1))


def binary_opacity_write_(opacities: Tensor, threshold: float) -> None:
    """``opacities.data = where(opacities >= thr, 1, 0)`` — in logit space, as the reference does."""
    # (written in place — the result is the reference's; the parameter may be a view of a flat slab)
    opacities.data.copy_(torch.where(opacities.data >= threshold, torch.ones_like(opacities.data),
                                     torch.zeros_like(opacities.data)))


def refine_schedule(cfg: SplatfactoConfig, step: int, num_train_data: int) -> Dict[str, bool]:
    """The three decisions of refinement_after (dn_model.py:328-341, 416-431) as pure host logic."""
    if step <= cfg.warmup_length:
        return dict(active=False, densify=False, cull_only=False, reset_opacity=False)
    reset_interval = cfg.reset_alpha_every * cfg.refine_every
    densify = step < cfg.stop_split_at and step % reset_interval > num_train_data + cfg.refine_every
    cull_only = (not densify) and step >= cfg.stop_split_at and cfg.continue_cull_post_densification
    reset = step < cfg.stop_split_at and step % reset_interval == cfg.refine_every
    return dict(active=True, densify=densify, cull_only=cull_only, reset_opacity=reset)


def densify_masks(cfg: SplatfactoConfig, step: int, xys_grad_norm: Tensor, vis_counts: Tensor,
                  max_2Dsize: Tensor, log_scales: Tensor, last_size, add_mask: Optional[Tensor]):
    """split / dup selection of dn_model.py:350-379.  NOTE the reference computes ``dups`` AFTER
    ``split_gaussians`` has divided the selected parents' scales by 1.6 in place (nerfstudio
    split_gaussians), so a parent just above the size threshold can be both
    split and duplicated; reproduced here.  Returns (splits, dups, scale_max_after)."""
    avg_grad_norm = (xys_grad_norm / vis_counts) * 0.5 * max(last_size[0], last_size[1])
    high_grads = avg_grad_norm > cfg.densify_grad_thresh
    scale_max = log_scales.exp().max(dim=-1).values
    splits = scale_max > cfg.densify_size_thresh
    if step < cfg.stop_screen_size_at:
        splits = splits | (max_2Dsize > cfg.split_screen_size)
    splits = splits & high_grads
    if add_mask is not None:
        splits = splits & ~add_mask
    shrunk = torch.log(torch.exp(log_scales) / 1.6).exp().max(dim=-1).values
    scale_max_after = torch.where(splits, shrunk, scale_max)
    dups = (scale_max_after <= cfg.densify_size_thresh) & high_grads
    if add_mask is not None:
        dups = dups & ~add_mask
    return splits, dups


def cull_mask(cfg: SplatfactoConfig, step: int, opacities: Tensor, log_scales: Tensor,
              max_2Dsize: Optional[Tensor], extra_cull_mask: Optional[Tensor]) -> Tensor:
    """cull_gaussians: which rows of the (grown) tensors are removed."""
    culls = (torch.sigmoid(opacities) < cfg.cull_alpha_thresh).squeeze(-1)
    if extra_cull_mask is not None:
        culls = culls | extra_cull_mask
    if step > cfg.refine_every * cfg.reset_alpha_every:
        toobigs = log_scales.exp().max(dim=-1).values > cfg.cull_scale_thresh
        if step < cfg.stop_screen_size_at and max_2Dsize is not None:
            toobigs = toobigs | (max_2Dsize > cfg.cull_screen_size)
        culls = culls | toobigs
    return culls


def _flush(trainer) -> None:
    """Callbacks that read or move parameter rows run between two steps; in a data-parallel run the previous step's
    deferred feature update (trainer.flush) must have landed before they look at anything."""
    if hasattr(trainer, "flush"):
        trainer.flush()


class DensifyStrategy:
    def __init__(self, cfg: Optional[SplatfactoConfig] = None, num_train_data: int = 1,
                 stats_only: bool = False):
        self.cfg = cfg or SplatfactoConfig()
        self.num_train_data = num_train_data
        self.stats_only = stats_only
        self.xys_grad_norm: Optional[Tensor] = None
        self.vis_counts: Optional[Tensor] = None
        self.max_2Dsize: Optional[Tensor] = None
        self.last_size = (1, 1)
        self.add_mask: Optional[Tensor] = None  # touch anchors (dn_model.py:292, 366-378)
        self.touch_normals: Optional[Tensor] = None  # [n_touch,3] contact normals of the anchors, patch order (:897-899)
        self._touch_rows = (None, None)
        # what the dataparser's metadata carries for the FusionSense callbacks (normal_nerfstudio.py:593-690)
        self.touch_patches = None
        self._staged_on = None
        self.gel_scale_factor = 6.34e-5   # normal_nerfstudio.py:72
        self.add_touch_at = 1000          # dn_model.py:136, configs/config.py:7
        self.visual_hull: Optional[Tensor] = None
        self.scale_factor = 1.0
        self.extra_cull_fn = None  # hook for hull / touch pruning (a-14)
        self.last_report: Dict[str, int] = {}

    def add_mask_u8(self) -> Optional[Tensor]:
        """The touch-anchor mask as the uint8 row the HIP kernels read, converted only when the mask object changed (the
        tape-free step converted it on every frame: one cast launch per step at BASELINE config #3)."""
        if self.add_mask is None:
            return None
        c = getattr(self, "_add_mask_u8", None)
        if c is None or c[0] is not self.add_mask:
            c = self._add_mask_u8 = (self.add_mask, self.add_mask.to(torch.uint8).contiguous())
        return c[1]

    def touch_rows(self) -> Optional[Tensor]:
        """Row indices of the touch anchors (``add_mask.nonzero()``), recomputed only when the mask object changed."""
        if self.add_mask is None:
            return None
        if self._touch_rows[0] is not self.add_mask:
            self._touch_rows = (self.add_mask, torch.nonzero(self.add_mask).squeeze(-1).contiguous())
        return self._touch_rows[1]

    # ---- a-12 ---------------------------------------------------------------------------
    @torch.no_grad()
    def _ensure_stats(self, N: int, dev) -> None:
        if self.xys_grad_norm is None:
            self.xys_grad_norm = torch.zeros(N, device=dev)
            self.vis_counts = torch.ones(N, device=dev)
        if self.max_2Dsize is None:
            self.max_2Dsize = torch.zeros(N, device=dev)

    def stats_target(self, trainer, N: int, dev, camera) -> Optional[dict]:
        """The statistics buffers for a backward that applies after_train itself (the fused get_outputs
        node does, in its per-Gaussian kernel); None when after_train would not run at this step."""
        if trainer.step >= self.cfg.stop_split_at:
            return None
        self._ensure_stats(N, dev)
        return {"xys_grad_norm": self.xys_grad_norm, "vis_counts": self.vis_counts, "max_2Dsize": self.max_2Dsize,
                "inv_max_hw": 1.0 / float(max(camera.height, camera.width))}

    def after_train(self, trainer, out, camera) -> None:
        if trainer.step >= self.cfg.stop_split_at:
            return
        radii = out["radii"]
        N = radii.shape[0]
        dev = radii.device
        self.last_size = (camera.height, camera.width)
        if getattr(out.get("info"), "stats_done", False):
            return  # already applied by the fused backward of this frame
        self._ensure_stats(N, dev)
        absgrad = out["xys"].absgrad[0]
        ops.densify_stats_(radii, absgrad, max(camera.height, camera.width), self.xys_grad_norm,
                           self.vis_counts, self.max_2Dsize)

    def _all_reduce_stats(self) -> None:
        """DP: every rank must take identical split/cull decisions."""
        from . import comm
        if comm.world() <= 1 or self.xys_grad_norm is None:
            return
        comm.all_reduce_(self.xys_grad_norm, "sum")
        extra = self.vis_counts - 1.0
        comm.all_reduce_(extra, "sum")
        self.vis_counts = extra + 1.0
        comm.all_reduce_(self.max_2Dsize, "max")

    def set_metadata(self, touch_patches=None, gel_scale_factor: float = 6.34e-5, add_touch_at: int = 1000,
                     visual_hull: Optional[Tensor] = None, scale_factor: float = 1.0) -> None:
        """The scene metadata the reference's callbacks read from ``self.kwargs["metadata"]``."""
        self.touch_patches, self.gel_scale_factor, self.add_touch_at = touch_patches, gel_scale_factor, add_touch_at
        self.visual_hull, self.scale_factor = visual_hull, scale_factor
        self._staged_on = None

    def stage(self, device) -> None:
        """The callbacks' scene metadata (touch patches, visual hull) moved to the training device ONCE — the reference
        keeps them on the host and copies them inside every callback (dn_model.py:1170-1190, 1262-1264): 93 small
        host-to-device copies with their waits were 9 of the 29 ms config #3's four callback steps took (round 5,
        tools/profile_refine.py).  Same values; called by SplatTrainer at construction and, lazily, by the callbacks."""
        if self._staged_on == str(device):
            return
        if self.touch_patches is not None:
            self.touch_patches = [{k: (v.to(device) if torch.is_tensor(v) else v) for k, v in p.items()}
                                  for p in self.touch_patches]
            from .touch import box_frames
            self._touch_frames = (self.touch_patches, box_frames(self.touch_patches, device))
        if self.visual_hull is not None:
            self.visual_hull = self.visual_hull.to(device)
        self._staged_on = str(device)
        if torch.device(device).type == "cuda":
            # everything the callbacks derive from the metadata alone, once: the anchors' rows but for their colours
            # (make_touch_gaussians: gel scales, +z -> contact-normal rotations), the hull's centre and its grid
            if self.touch_patches is not None:
                self._touch_static_rows(self.touch_patches, self.gel_scale_factor, torch.device(device))
            if self.visual_hull is not None:
                from .touch import hull_grid
                self._hull_center(self.visual_hull)
                self._hull_grid_of = (self.visual_hull, float(self.scale_factor),
                                      hull_grid(self.visual_hull, 0.02 * self.scale_factor))

    def _frames_of(self, touch_patches):
        """The staged boxes' frames when ``touch_patches`` is the staged list (else None: touch_aabb_mask derives them)."""
        tf = getattr(self, "_touch_frames", None)
        return tf[1] if (tf is not None and tf[0] is touch_patches) else None

    def before_train(self, trainer) -> None:
        """BEFORE_TRAIN_ITERATION callbacks (dn_model.py:1370-1383): add_touch_patch at step == add_touch_at."""
        if self.stats_only or self.touch_patches is None or trainer.step != self.add_touch_at:
            return
        self.add_touch_patch(trainer, self.touch_patches, self.gel_scale_factor)

    def maybe_refine(self, trainer) -> None:
        """AFTER_TRAIN_ITERATION callbacks with update_every_num_iters=refine_every, in the reference's order
        (dn_model.py:1394-1424): refinement_after, hull_pruning, touch_pruning."""
        if self.stats_only:
            return
        if trainer.step % self.cfg.refine_every == 0:
            if hasattr(trainer, "flush"):
                trainer.flush()  # a deferred feature update must land before rows are split / culled
            self.refinement_after(trainer, trainer.step)
            if self.visual_hull is not None and self.touch_patches is not None:
                self.hull_and_touch_pruning(trainer)
            elif self.visual_hull is not None:
                self.hull_pruning(trainer, self.visual_hull, self.scale_factor)
            elif self.touch_patches is not None:
                self.touch_pruning(trainer, self.touch_patches)

    # ---- a-13 ---------------------------------------------------------------------------
    def _rules(self, step: int, densify: bool):
        """The schedule scalars and thresholds of one refinement / cull for fsgs_refine_mark (thresholds go over as fp32:
        what a float32 tensor compared with a Python scalar sees in the reference)."""
        from ._lib import RefineRules
        cfg = self.cfg
        r = RefineRules()
        r.densify, r.n_samples = (1 if densify else 0), int(cfg.n_split_samples)
        r.max_hw = float(max(self.last_size[0], self.last_size[1]))
        r.densify_grad_thresh, r.densify_size_thresh = cfg.densify_grad_thresh, cfg.densify_size_thresh
        r.check_screen, r.split_screen_size = int(step < cfg.stop_screen_size_at), cfg.split_screen_size
        r.cull_alpha_thresh = cfg.cull_alpha_thresh
        r.check_big = int(step > cfg.refine_every * cfg.reset_alpha_every)
        r.cull_scale_thresh, r.cull_screen_size = cfg.cull_scale_thresh, cfg.cull_screen_size
        r.hull_close, r.hull_lo, r.hull_hi, r.n_boxes = -1.0, 0.0, 0.0, 0
        return r

    @torch.no_grad()
    def refinement_after(self, trainer, step: int) -> None:
        """dn_model.py:326-451 in two library launches around one host read (csrc/refine.hip): fsgs_refine_mark takes every
        decision (densify_masks / cull_mask above state them in torch; tests/test_gpu_densify.py holds the kernel to them),
        the host learns five totals, draws the split noise — ``torch.randn((samps * n_split, 3))`` from the trainer's
        generator, as before — and fsgs_refine_move writes the new parameter and moment tensors in one pass."""
        cfg = self.cfg
        sched = refine_schedule(cfg, step, self.num_train_data)
        if not sched["active"]:
            return
        self._all_reduce_stats()
        P = trainer.params
        N = P["means"].shape[0]
        dev = P["means"].device
        n_split = n_dup = deleted = 0
        if sched["densify"] or sched["cull_only"]:
            densify = sched["densify"]
            if densify:
                assert self.xys_grad_norm is not None and self.vis_counts is not None and self.max_2Dsize is not None
            extra = self.extra_cull_fn(trainer) if self.extra_cull_fn is not None else None
            marks = ops.refine_mark(self._rules(step, densify), P["scales"].data, P["opacities"].data,
                                    self.xys_grad_norm if densify else None, self.vis_counts if densify else None,
                                    self.max_2Dsize, add_mask=self.add_mask, extra_cull=extra)
            randn = None
            if densify:
                n_split, n_dup = marks.n_split, marks.n_dup
                randn = torch.randn((cfg.n_split_samples * n_split, 3), device=dev, generator=trainer.rng)
            n_new = marks.n_samples * marks.n_split + n_dup
            deleted = (N - marks.n_keep) + (n_new - marks.n_samples * marks.n_split_kept - marks.n_dup_kept)
            self._rebuild_marked(trainer, marks, randn)

        if sched["reset_opacity"]:
            reset_value = cfg.cull_alpha_thresh * 2.0
            logit = torch.logit(torch.tensor(reset_value)).item()
            if hasattr(trainer, "sync_optimizer_state"):
                trainer.sync_optimizer_state()
            if hasattr(trainer, "drop_occlusion_cuts"):
                trainer.drop_occlusion_cuts()  # (nothing saturates where it did any more)
            op = trainer.params["opacities"]
            op.data.clamp_(max=logit)  # (in place: the parameter may be a view of the trainer's flat geometry slab)
            if hasattr(trainer, "mark_params_written"):
                trainer.mark_params_written()
            st = trainer.optimizers["opacities"].state.get(op, None)
            if st:
                st["exp_avg"].zero_()
                st["exp_avg_sq"].zero_()

        self.xys_grad_norm = None
        self.vis_counts = None
        self.max_2Dsize = None
        self.last_report = dict(step=step, n_before=N, n_split=n_split, n_dup=n_dup, n_deleted=deleted,
                                n_after=trainer.params["means"].shape[0])

    # ---- a-14: FusionSense-specific callbacks (dn_model.py:1156-1302) ----------------------
    @torch.no_grad()
    def cull_gaussians(self, trainer, extra_cull_mask: Optional[Tensor] = None, hull=None, boxes: Optional[Tensor] = None,
                       exempt_anchors: bool = True, append_rows: Optional[Dict[str, Tensor]] = None) -> Tensor:
        """SplatfactoModel.cull_gaussians + remove_from_all_optim in one pass (mark, five totals, move).  Returns the
        deleted mask over the rows that existed before the call.
        The pruning callbacks' own tests ride in the mark: ``hull`` = (words of ops.nearest_point_words or None, the hull's
        grid or None, centre, scale factor) -> hull_pruning's shell test, ``boxes`` = the patches' frames -> touch_pruning's box test; touch anchors
        pass both unless ``exempt_anchors`` is off (add_touch_patch culls whatever lies inside the boxes).
        ``append_rows``: rows added behind the survivors in the same move, as touch anchors (add_touch_patch)."""
        _flush(trainer)
        P = trainer.params
        rules = self._rules(trainer.step, False)
        words = grid = None
        if hull is not None:
            words, grid, center, scale_factor = hull
            rules.hull_center[0], rules.hull_center[1], rules.hull_center[2] = center
            rules.hull_close, rules.hull_lo, rules.hull_hi = 0.2 * scale_factor, 0.005 * scale_factor, 0.02 * scale_factor
        marks = ops.refine_mark(rules, P["scales"].data, P["opacities"].data, max_2Dsize=self.max_2Dsize,
                                means=P["means"].data, add_mask=self.add_mask if exempt_anchors else None,
                                extra_cull=extra_cull_mask, hull_words=words, boxes=boxes, hull_grid=grid)
        deleted = marks.deleted()
        self._rebuild_marked(trainer, marks, None, append_rows=append_rows, stats=append_rows is None)
        return deleted

    @torch.no_grad()
    def append_gaussians(self, trainer, rows: Dict[str, Tensor]) -> None:
        """Append rows with zero Adam moments (add_in_all_optim, dn_model.py:1150-1152)."""
        _flush(trainer)
        P = trainer.params
        n_old = P["means"].shape[0]
        marks = ops.RefineMarks()  # every row stays: no mark launch, the offsets are the rows themselves
        marks.n, marks.n_keep, marks.n_split, marks.n_dup, marks.n_split_kept, marks.n_dup_kept, marks.n_samples = \
            n_old, n_old, 0, 0, 0, 0, 0
        marks.flags, marks.book = ops.refine_keep_all(n_old, P["means"].device)
        self._rebuild_marked(trainer, marks, None, append_rows=rows, anchors=False)

    def _touch_static_rows(self, touch_patches, gel_scale_factor: float, device):
        """(points, normals, the rows make_touch_gaussians builds from them) of a patch list — everything but the colours
        is a function of the patches alone (dn_model.py:1190-1224: gel scales, the +z -> contact-normal rotations, raw
        opacity 1), so it is built once per list and kept."""
        from .touch import make_touch_gaussians
        c = getattr(self, "_touch_static", None)
        key = (float(gel_scale_factor), int(self.cfg.sh_degree), str(device))
        if c is not None and c[0] is touch_patches and c[1] == key:
            return c[2]
        pts = [p["points_xyz"].to(device) for p in touch_patches if p["points_xyz"].shape[0] > 0]
        if not pts:
            out = (None, None, None)
        else:
            pts = torch.cat(pts).to(torch.float32).contiguous()
            nrm = torch.cat([p["normals"].to(device) for p in touch_patches if p["points_xyz"].shape[0] > 0])
            rows = make_touch_gaussians(pts, nrm, torch.full_like(pts, 0.5), gel_scale_factor, self.cfg.sh_degree)
            out = (pts, nrm, rows)
        self._touch_static = (touch_patches, key, out)
        return out

    @torch.no_grad()
    def add_touch_patch(self, trainer, touch_patches, gel_scale_factor: float, colors_fn=None) -> int:
        """dn_model.py:1156-1247 at step == add_touch_at: cull Gaussians inside the patches'
        oriented boxes, then append the touch points as fixed anchors (``add_mask``).  Colours
        of the new points come from the nearest existing Gaussian (knn_sk, k=1, dn_model.py:1181-1182).
        One nearest-point search for all patches' points, then ONE mark + move: the box test rides in the mark, the
        anchors are appended behind the survivors by the move."""
        _flush(trainer)
        from .touch import box_frames
        from .scenes import rgb_to_sh, sh_to_rgb
        P = trainer.params
        means = P["means"].data
        dev = means.device
        if not means.is_cuda:
            raise ValueError("add_touch_patch runs on the GPU (fsgs_nearest_point, fsgs_refine_mark): there is no CPU path")
        if touch_patches is self.touch_patches:
            self.stage(dev)
            touch_patches = self.touch_patches
        frames = self._frames_of(touch_patches)
        if frames is None:
            frames = box_frames(touch_patches, dev)
        pts, nrm, static = self._touch_static_rows(touch_patches, gel_scale_factor, dev)
        if pts is None:
            self.cull_gaussians(trainer, boxes=frames, exempt_anchors=False)
            return 0
        # 1-NN on the device (reference: sklearn on CPU); brute force in libfsgs, no BLAS, no [n, N] matrix
        nn_idx = ops.nearest_point(pts, means, want_idx=True)[1]
        near = P["features_dc"].data[nn_idx]
        rgb = sh_to_rgb(near) if self.cfg.sh_degree > 0 else torch.sigmoid(near)
        rows = dict(static)
        rows["features_dc"] = rgb_to_sh(rgb) if self.cfg.sh_degree > 0 else torch.logit(rgb, eps=1e-10)
        self.add_mask = None  # (the reference starts the mask over: zeros for every row that was there, ones for the anchors)
        self.cull_gaussians(trainer, boxes=frames, exempt_anchors=False, append_rows=rows)
        self.touch_normals = nrm.to(torch.float32).contiguous()
        self.xys_grad_norm = self.vis_counts = self.max_2Dsize = None
        return pts.shape[0]

    def _hull_center(self, visual_hull: Tensor):
        """visual_hull.mean(0) as three Python floats, kept per hull tensor (hull_pruning recomputed it every call)."""
        c = getattr(self, "_hull_center_of", None)
        if c is None or c[0] is not visual_hull:
            c = self._hull_center_of = (visual_hull, [float(x) for x in visual_hull.mean(dim=0).tolist()])
        return c[1]

    def _hull_words(self, trainer, visual_hull: Tensor, scale_factor: float):
        """What cull_gaussians needs for hull_pruning's test: the hull's grid (built once per hull and scale factor: the
        search then rides in the mark launch) or, for a hull too large for a grid, the exhaustive search's words."""
        means = trainer.params["means"].data
        if not means.is_cuda:
            raise ValueError("hull_pruning runs on the GPU (fsgs_refine_mark): there is no CPU path")
        center = self._hull_center(visual_hull)
        c = getattr(self, "_hull_grid_of", None)
        if c is None or c[0] is not visual_hull or c[1] != float(scale_factor):
            from .touch import hull_grid
            c = self._hull_grid_of = (visual_hull, float(scale_factor), hull_grid(visual_hull, 0.02 * scale_factor))
        if c[2] is not None and os.environ.get("FSGS_HULL_GRID", "1") != "0":
            return None, c[2], center, scale_factor
        return ops.nearest_point_words(means, visual_hull, center, 0.2 * scale_factor), None, center, scale_factor

    @torch.no_grad()
    def hull_pruning(self, trainer, visual_hull: Tensor, scale_factor: float) -> Optional[Tensor]:
        """dn_model.py:1249-1276: Gaussians near the object (<= 0.2 s from the hull centre) whose distance to the hull
        point set is in (0.005 s, 0.02 s] are culled; touch anchors never."""
        _flush(trainer)
        if trainer.step <= self.cfg.warmup_length:
            return None
        if visual_hull is self.visual_hull:
            self.stage(trainer.device)
            visual_hull = self.visual_hull
        elif visual_hull.device != trainer.params["means"].device:
            visual_hull = visual_hull.to(trainer.params["means"].device)
        self.max_2Dsize = None
        return self.cull_gaussians(trainer, hull=self._hull_words(trainer, visual_hull, scale_factor))

    def _frames(self, trainer, touch_patches) -> Tensor:
        from .touch import box_frames
        if touch_patches is self.touch_patches:
            self.stage(trainer.device)
            touch_patches = self.touch_patches
        frames = self._frames_of(touch_patches)
        return frames if frames is not None else box_frames(touch_patches, trainer.params["means"].device)

    @torch.no_grad()
    def touch_pruning(self, trainer, touch_patches) -> Optional[Tensor]:
        """dn_model.py:1279-1302."""
        _flush(trainer)
        if trainer.step <= self.cfg.warmup_length or self.add_mask is None:
            return None
        return self.cull_gaussians(trainer, boxes=self._frames(trainer, touch_patches))

    @torch.no_grad()
    def hull_and_touch_pruning(self, trainer) -> Optional[Tensor]:
        """hull_pruning followed by touch_pruning (the reference's order, dn_model.py:1394-1424) with ONE cull and one
        rebuild of the parameters instead of two: every test of the two callbacks is a function of the row alone —
        sigmoid(opacity) < cull_alpha_thresh, the too-big test on the row's own scales (no screen sizes: both callbacks
        run with max_2Dsize = None), the distance to the hull, the boxes of the patches — so the rows that survive
        "cull A, then cull B on the survivors" are the rows that fail neither test.  Same survivors, same order
        (tests/test_gpu_densify.py checks it against the two calls); returns the deleted mask over the rows before it."""
        if trainer.step <= self.cfg.warmup_length:
            return None
        if self.add_mask is None:  # touch_pruning does nothing before add_touch_patch
            return self.hull_pruning(trainer, self.visual_hull, self.scale_factor)
        _flush(trainer)
        self.stage(trainer.device)
        self.max_2Dsize = None
        return self.cull_gaussians(trainer, hull=self._hull_words(trainer, self.visual_hull, self.scale_factor),
                                   boxes=self._frames(trainer, self.touch_patches))

    def _rebuild_marked(self, trainer, marks, randn: Optional[Tensor], append_rows: Optional[Dict[str, Tensor]] = None,
                        stats: bool = False, anchors: bool = True) -> None:
        """New parameter tensors + Adam moments from the marks of fsgs_refine_mark, in ONE launch (fsgs_refine_move): kept
        old rows (order preserved), the kept split children sample by sample, the kept duplicates, then ``append_rows``;
        new rows get zero moments (dup_in_optim), culled rows drop theirs (remove_from_optim).  ``stats``: the after_train
        statistics are compacted with the rows (a cull between two refinements).  ``add_mask`` follows the rows: zeros
        for children / duplicates, ones for appended rows when ``anchors``."""
        if hasattr(trainer, "flush"):
            # a deferred feature update (data-parallel runs) is sized for the old N and reads the old slab: it must
            # land before any row moves — add_touch_patch / hull_pruning / touch_pruning run between two steps
            trainer.flush()
        if hasattr(trainer, "sync_optimizer_state"):
            trainer.sync_optimizer_state()  # (sharded geometry step: whole moment tensors are about to be compacted)
        # (per-view occlusion cuts are depths per tile, not rows: they outlive the rebuild — a split or cull moves a
        # tile's saturation depth little, and a cut that has become too tight is found out on the device; measured over
        # 1 500 steps with a refinement every 50: 1 482 cut frames against 1 356 when every rebuild dropped them, none
        # redone either way.  An opacity reset does drop them: nothing saturates where it did.)
        P = trainer.params
        dev = P["means"].device
        n_append = int(append_rows["means"].shape[0]) if append_rows else 0
        n_final = marks.n_keep + marks.n_samples * marks.n_split_kept + marks.n_dup_kept + n_append
        kinds = {"means": ops.COL_MEANS, "scales": ops.COL_SCALES}
        cols, fresh = [], {}
        for name in list(P.keys()):
            old_p = P[name]
            opt = trainer.optimizers[name]
            shape = (n_final,) + tuple(old_p.shape[1:])
            new_data = torch.empty(shape, dtype=old_p.dtype, device=dev)
            cols.append((old_p.data, new_data, append_rows[name].reshape((n_append,) + tuple(old_p.shape[1:]))
                         if append_rows else None, kinds.get(name, ops.COL_COPY)))
            state = opt.state.pop(old_p, None)
            moments = {}
            if state:
                for key in ("exp_avg", "exp_avg_sq"):
                    buf = torch.empty(shape, dtype=old_p.dtype, device=dev)
                    cols.append((state[key], buf, None, ops.COL_ZERO))
                    moments[key] = buf
            fresh[name] = (new_data, state, moments)
        moved_stats = {}
        if stats:
            for name in ("xys_grad_norm", "vis_counts", "max_2Dsize"):
                t = getattr(self, name)
                if t is not None:
                    moved_stats[name] = torch.empty(n_final, dtype=torch.float32, device=dev)
                    cols.append((t, moved_stats[name], None, ops.COL_ZERO))
        mask_in = mask_out = None
        if self.add_mask is not None or (n_append > 0 and anchors):
            mask_in = self.add_mask if self.add_mask is not None else torch.zeros(marks.n, dtype=torch.bool, device=dev)
            mask_out = torch.empty(n_final, dtype=torch.bool, device=dev)
        ops.refine_move(marks, cols, n_append, randn, P["quats"].data, P["scales"].data, mask_in, mask_out,
                        1 if anchors else 0)
        for name, (new_data, state, moments) in fresh.items():
            opt = trainer.optimizers[name]
            new_p = torch.nn.Parameter(new_data)
            if state:
                state.update(moments)
                opt.state[new_p] = state
            opt.param_groups[0]["params"] = [new_p]
            trainer.params[name] = new_p
        for name, t in moved_stats.items():
            setattr(self, name, t)
        if mask_out is not None:
            self.add_mask = mask_out
        trainer.slab.rebuild(trainer.params)
        if hasattr(trainer, "mark_params_written"):
            trainer.mark_params_written()
