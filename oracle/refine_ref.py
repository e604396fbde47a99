"""CPU oracle for csrc/refine.hip (TEST INFRASTRUCTURE ONLY; parity unpinned — see oracle/gsplat_ref.py): the two
launches fsgs_refine_mark / fsgs_refine_move stated in plain torch, the reference's way — boolean masks, ``torch.cat``
of old and new rows, boolean indexing — so that the kernels can be held to them flag for flag and row for row, and so that
the CPU (gloo) tests of the data-parallel refinement have row primitives to run on.

What is restated: ``DNSplatterModel.refinement_after`` (/root/reference/dn_splatter/dn_model.py:326-451: split / duplicate
selection :342-379, the cull :403-431), nerfstudio 1.1.3 ``split_gaussians`` / ``dup_gaussians`` / ``cull_gaussians`` as
FusionSense uses them, ``dup_in_all_optim`` / ``remove_from_all_optim`` (:1120-1152: new rows get
zero Adam moments), and the row tests of ``hull_pruning`` / ``touch_pruning`` (:1249-1302).  oracle/splatfacto_ref.py
states the same refinement end to end; this file states it cut at the kernels' boundary (flags, then rows)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

KEEP, SPLIT, DUP, CHILD_KEEP, DUP_KEEP = 1, 2, 4, 8, 16
COL_COPY, COL_ZERO, COL_MEANS, COL_SCALES = 0, 1, 2, 3


class Marks:
    pass


def hull_distance_words(means: Tensor, hull: Tensor, center, close_radius: float) -> Tensor:
    """fsgs_nearest_point_words: squared distance to the nearest hull point (fp32 bits << 32 | index) for the rows within
    ``close_radius`` of ``center``, all ones elsewhere."""
    c = torch.tensor(center, dtype=torch.float32)
    close = (means - c).pow(2).sum(-1).sqrt() <= torch.tensor(close_radius, dtype=torch.float32)
    d2 = ((means[:, None, :] - hull[None, :, :]) ** 2).sum(-1)
    best, idx = d2.min(dim=-1)
    words = (best.contiguous().view(torch.int32).to(torch.int64) << 32) | idx
    return torch.where(close, words, torch.full_like(words, -1))


def in_boxes(means: Tensor, boxes: Tensor) -> Tensor:
    """Inside any oriented box (18 floats: centre, three axes, lower and upper bounds; dn_model.py:1996-2034)."""
    out = torch.zeros(means.shape[0], dtype=torch.bool)
    for b in boxes:
        rel = (means - b[0:3]) @ b[3:12].reshape(3, 3).T
        out |= ((rel >= b[12:15]) & (rel <= b[15:18])).all(dim=-1)
    return out


def mark(rules, log_scales: Tensor, opac_logit: Tensor, xys_grad_norm: Optional[Tensor] = None,
         vis_counts: Optional[Tensor] = None, max_2Dsize: Optional[Tensor] = None, means: Optional[Tensor] = None,
         add_mask: Optional[Tensor] = None, extra_cull: Optional[Tensor] = None, hull_words: Optional[Tensor] = None,
         boxes: Optional[Tensor] = None) -> Marks:
    """The flag byte per row and the five totals (``rules``: an object with fsgs_refine_rules' fields)."""
    n = log_scales.shape[0]
    f32 = lambda x: torch.tensor(float(x), dtype=torch.float32)  # noqa: E731  (thresholds as the kernels receive them)
    e = torch.exp(log_scales)
    smax = e.max(dim=-1).values if n else torch.zeros(0)
    cmax = torch.log(e / 1.6).exp().max(dim=-1).values if n else torch.zeros(0)  # split_gaussians' children, seen by the cull
    low = (torch.sigmoid(opac_logit.reshape(n)) < f32(rules.cull_alpha_thresh))
    add = add_mask.bool() if add_mask is not None else torch.zeros(n, dtype=torch.bool)
    culled = low.clone()
    if extra_cull is not None:
        culled |= extra_cull.bool()
    zeros = torch.zeros(n, dtype=torch.bool)
    split, dup, child_keep, dup_keep = zeros, zeros, zeros, zeros
    big_on = bool(rules.check_big)
    screen_on = bool(rules.check_screen) and max_2Dsize is not None
    if rules.densify:
        avg = (xys_grad_norm / vis_counts) * 0.5 * f32(rules.max_hw)          # dn_model.py:352-353
        high = avg > f32(rules.densify_grad_thresh)
        split = smax > f32(rules.densify_size_thresh)                           # :354-358
        if screen_on:
            split = split | (max_2Dsize > f32(rules.split_screen_size))
        split = split & high & ~add                                             # :359, :366-368 (anchors never)
        after = torch.where(split, cmax, smax)       # the duplicates are chosen after split_gaussians' in-place /1.6
        dup = (after <= f32(rules.densify_size_thresh)) & high & ~add           # :371-378
        culled |= split                                                         # :403-415: a split parent goes
        child_keep = split & ~(low | (big_on & (cmax > f32(rules.cull_scale_thresh))))
        dup_keep = dup & ~(low | (big_on & (after > f32(rules.cull_scale_thresh))))
    if big_on:
        big = smax > f32(rules.cull_scale_thresh)
        if screen_on:
            big = big | (max_2Dsize > f32(rules.cull_screen_size))
        culled |= big
    if hull_words is not None:                                                  # :1249-1276
        c = torch.tensor(list(rules.hull_center), dtype=torch.float32)
        close = (means - c).pow(2).sum(-1).sqrt() <= f32(rules.hull_close)
        none = hull_words == -1
        d = torch.where(none, torch.full((n,), float("inf")),
                        (hull_words >> 32).to(torch.int32).view(torch.float32).sqrt())
        culled |= close & (d > f32(rules.hull_lo)) & (d <= f32(rules.hull_hi)) & ~add
    if boxes is not None and boxes.shape[0] > 0:                                # :1279-1302
        culled |= in_boxes(means, boxes) & ~add
    m = Marks()
    m.n = n
    m.flags = ((~culled).to(torch.uint8) * KEEP + split.to(torch.uint8) * SPLIT + dup.to(torch.uint8) * DUP
               + child_keep.to(torch.uint8) * CHILD_KEEP + dup_keep.to(torch.uint8) * DUP_KEEP)
    m.book = None
    m.n_keep, m.n_split, m.n_dup = int((~culled).sum()), int(split.sum()), int(dup.sum())
    m.n_split_kept, m.n_dup_kept = int(child_keep.sum()), int(dup_keep.sum())
    m.n_samples = int(rules.n_samples) if rules.densify else 0
    m.deleted = lambda: culled.clone()
    return m


def _rotmat(q: Tensor) -> Tensor:
    q = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)


def move(marks: Marks, cols, n_append: int = 0, randn: Optional[Tensor] = None, quats: Optional[Tensor] = None,
         log_scales: Optional[Tensor] = None, mask_in: Optional[Tensor] = None, mask_out: Optional[Tensor] = None,
         mask_append: int = 0) -> None:
    """``cols`` = [(src, dst, append rows or None, kind)]: dst <- cat([kept old rows, the kept children sample by sample,
    the kept duplicates, the appended rows]) with what a NEW row holds decided by the column's kind."""
    f = marks.flags[:marks.n]
    keep = (f & KEEP) != 0
    split = (f & SPLIT) != 0
    child = split & ((f & CHILD_KEEP) != 0)
    dupk = ((f & DUP) != 0) & ((f & DUP_KEEP) != 0)
    samps = marks.n_samples
    rank_in_split = torch.cumsum(split.to(torch.int64), 0) - 1       # the parent's rank among ALL split rows
    for src, dst, app, kind in cols:
        parts = [src[keep]]
        for s in range(samps):
            rows = src[child]
            if kind == COL_ZERO:
                rows = torch.zeros_like(rows)
            elif kind == COL_SCALES:
                rows = torch.log(torch.exp(rows) / 1.6)
            elif kind == COL_MEANS:
                z = randn[s * marks.n_split + rank_in_split[child]]
                rows = (_rotmat(quats[child]) @ (torch.exp(log_scales[child]) * z)[..., None]).squeeze(-1) + rows
            parts.append(rows)
        rows = src[dupk]
        if kind == COL_ZERO:
            rows = torch.zeros_like(rows)
        elif kind == COL_SCALES:
            rows = torch.where(split[dupk].reshape((-1,) + (1,) * (rows.dim() - 1)), torch.log(torch.exp(rows) / 1.6), rows)
        parts.append(rows)
        if n_append:
            parts.append(torch.zeros((n_append,) + tuple(src.shape[1:])) if kind == COL_ZERO
                         else app.to(torch.float32).reshape((n_append,) + tuple(src.shape[1:])))
        dst.copy_(torch.cat(parts, dim=0))
    if mask_out is not None:
        n_new = samps * marks.n_split_kept + marks.n_dup_kept
        mask_out.copy_(torch.cat([mask_in[keep].bool(), torch.zeros(n_new, dtype=torch.bool),
                                  torch.full((n_append,), bool(mask_append))]))
