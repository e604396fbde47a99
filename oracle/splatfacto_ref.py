"""CPU oracle for the densify/prune path (TEST INFRASTRUCTURE ONLY; parity unpinned — see
oracle/gsplat_ref.py).  Plain-torch restatement of nerfstudio 1.1.3 ``SplatfactoModel``
methods as FusionSense uses them and of
``DNSplatterModel.refinement_after`` (/root/reference/dn_splatter/dn_model.py:326-451),
written the reference's way: torch.cat of all tensors followed by boolean indexing."""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from .gsplat_ref import quat_to_rotmat


def after_train(state: Dict, radii: Tensor, absgrad: Tensor, H: int, W: int, step: int, stop_split_at: int):
    """state: dict with xys_grad_norm / vis_counts / max_2Dsize (None initially)."""
    if step >= stop_split_at:
        return
    vis = (radii > 0).flatten()
    g = absgrad[vis].norm(dim=-1)
    N = radii.numel()
    if state.get("xys_grad_norm") is None:
        state["xys_grad_norm"] = torch.zeros(N)
        state["vis_counts"] = torch.ones(N)
    state["vis_counts"][vis] += 1
    state["xys_grad_norm"][vis] += g
    if state.get("max_2Dsize") is None:
        state["max_2Dsize"] = torch.zeros(N)
    newradii = radii.flatten()[vis]
    state["max_2Dsize"][vis] = torch.maximum(state["max_2Dsize"][vis], newradii / float(max(H, W)))


def refinement_after(params: Dict[str, Tensor], adam: Dict[str, Dict[str, Tensor]], state: Dict, cfg,
                     step: int, num_train_data: int, last_size, randn: Optional[Tensor] = None,
                     add_mask: Optional[Tensor] = None):
    """Returns (new_params, new_adam, report).  ``randn`` [samps*n_split,3] supplies the split
    noise so a GPU run with the same numbers can be compared exactly."""
    report = dict(n_split=0, n_dup=0, n_deleted=0)
    if step <= cfg.warmup_length:
        return params, adam, report
    params = {k: v.clone() for k, v in params.items()}
    adam = {k: {kk: vv.clone() for kk, vv in v.items()} for k, v in adam.items()}
    reset_interval = cfg.reset_alpha_every * cfg.refine_every
    do_densification = step < cfg.stop_split_at and step % reset_interval > num_train_data + cfg.refine_every
    deleted = None
    max_2Dsize = state.get("max_2Dsize")

    def cull(extra):
        nonlocal params, max_2Dsize
        culls = (torch.sigmoid(params["opacities"]) < cfg.cull_alpha_thresh).squeeze(-1)
        if extra is not None:
            culls = culls | extra
        if step > cfg.refine_every * cfg.reset_alpha_every:
            toobigs = torch.exp(params["scales"]).max(dim=-1).values > cfg.cull_scale_thresh
            if step < cfg.stop_screen_size_at and max_2Dsize is not None:
                toobigs = toobigs | (max_2Dsize > cfg.cull_screen_size)
            culls = culls | toobigs
        params = {k: v[~culls] for k, v in params.items()}
        return culls

    if do_densification:
        avg = (state["xys_grad_norm"] / state["vis_counts"]) * 0.5 * max(last_size[0], last_size[1])
        high = avg > cfg.densify_grad_thresh
        splits = torch.exp(params["scales"]).max(dim=-1).values > cfg.densify_size_thresh
        if step < cfg.stop_screen_size_at:
            splits = splits | (max_2Dsize > cfg.split_screen_size)
        splits = splits & high
        if add_mask is not None:
            splits = splits & ~add_mask
        samps = cfg.n_split_samples
        n_splits = int(splits.sum())
        # split_gaussians
        z = randn if randn is not None else torch.randn(samps * n_splits, 3)
        scaled = torch.exp(params["scales"][splits].repeat(samps, 1)) * z
        q = params["quats"][splits] / params["quats"][splits].norm(dim=-1, keepdim=True)
        rots = quat_to_rotmat(q.repeat(samps, 1))
        new_means = torch.bmm(rots, scaled[..., None]).squeeze(-1) + params["means"][splits].repeat(samps, 1)
        new_scales = torch.log(torch.exp(params["scales"][splits]) / 1.6).repeat(samps, 1)
        params["scales"][splits] = torch.log(torch.exp(params["scales"][splits]) / 1.6)  # in place (A.2)
        split_params = {"means": new_means, "scales": new_scales}
        for k, v in params.items():
            if k not in split_params:
                split_params[k] = v[splits].repeat(samps, *([1] * (v.dim() - 1)))
        dups = torch.exp(params["scales"]).max(dim=-1).values <= cfg.densify_size_thresh
        dups = dups & high
        if add_mask is not None:
            dups = dups & ~add_mask
        dup_params = {k: v[dups] for k, v in params.items()}
        n_dups = int(dups.sum())
        params = {k: torch.cat([v, split_params[k], dup_params[k]], dim=0) for k, v in params.items()}
        max_2Dsize = torch.cat([max_2Dsize, torch.zeros(samps * n_splits), torch.zeros(n_dups)])
        for k in adam:  # dup_in_optim twice: zeros appended
            for m in ("exp_avg", "exp_avg_sq"):
                t = adam[k][m]
                adam[k][m] = torch.cat([t, torch.zeros((samps * n_splits + n_dups,) + t.shape[1:])], dim=0)
        splits_mask = torch.cat([splits, torch.zeros(samps * n_splits + n_dups, dtype=torch.bool)])
        deleted = cull(splits_mask)
        report.update(n_split=n_splits, n_dup=n_dups)
    elif step >= cfg.stop_split_at and cfg.continue_cull_post_densification:
        deleted = cull(None)
    if deleted is not None:
        for k in adam:
            for m in ("exp_avg", "exp_avg_sq"):
                adam[k][m] = adam[k][m][~deleted]
        report["n_deleted"] = int(deleted.sum())
    if step < cfg.stop_split_at and step % reset_interval == cfg.refine_every:
        reset_value = cfg.cull_alpha_thresh * 2.0
        params["opacities"] = torch.clamp(params["opacities"], max=torch.logit(torch.tensor(reset_value)).item())
        adam["opacities"]["exp_avg"] = torch.zeros_like(adam["opacities"]["exp_avg"])
        adam["opacities"]["exp_avg_sq"] = torch.zeros_like(adam["opacities"]["exp_avg_sq"])
    state["xys_grad_norm"] = state["vis_counts"] = state["max_2Dsize"] = None
    return params, adam, report
