"""The SuGaR-style SDF term of FusionSense's training loss (``use_sdf_loss``; off in FusionSense's configuration:
/root/reference/dn_splatter/dn_model.py:102), restated from the reference's own methods — line for line in MEANING,
quirks included, because the term is pinned against the reference executing them (tests/golden/reference_loss.npz,
case ``sdf``):

* ``sample_points_in_gaussians`` (:1445-1495) draws which Gaussians to sample with ``torch.multinomial`` over the
  CUMULATIVE volume fractions (not the fractions), then one standard-normal offset per sample in the Gaussian's frame;
* ``get_ideal_sdf`` (:1497-1546) projects the samples with ``fy = fx`` and keeps pixels strictly inside (0, W) x (0, H);
* ``get_density`` (:1575-1636) sums the k tracked neighbours' opacity-weighted Gaussians, renormalises densities >= 1 by
  their own detached value + 1e-5 and clamps at 1e-4; ``get_sdf`` (:1638-1660) is sqrt(-2 log density);
* ``get_sdf_loss_weight`` ("std", :1668-1700) indexes the GAUSSIANS with the SAMPLES' validity mask — the reference
  only runs when there are as many samples as Gaussians (``num_points > num_sdf_samples``: then num_samples =
  num_points), and so does this.

Plain torch on whatever device the parameters live on: a dozen small launches on the steps that have the term (every
``apply_sdf_loss_iters``-th step after ``apply_sdf_loss_after_iters``), autograd carries the gradients — like the other
terms that ``integration._extra_terms`` adds on top of the fused loss node.  The random draws can be handed in
(``draws``) so that a test can replay the reference's own."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from .legacy import quat_to_rotmat
from .touch import invert_quaternion


def sample_points_in_gaussians(means: Tensor, scales: Tensor, quats: Tensor, num_samples: int,
                               vis_indices: Optional[Tensor] = None, draws: Optional[Tuple[Tensor, Tensor]] = None,
                               generator: Optional[torch.Generator] = None) -> Tuple[Tensor, Tensor]:
    """dn_model.py:1445-1495.  ``draws`` = (multinomial picks [num_samples] int64, normal offsets [num_samples, 3])."""
    vis_scales = torch.exp(scales[vis_indices]) if vis_indices is not None else torch.exp(scales)
    areas = (vis_scales[..., 0] * vis_scales[..., 1] * vis_scales[..., 2]).abs()
    cum_probs = areas.cumsum(dim=-1) / areas.sum(dim=-1, keepdim=True)
    if draws is not None:
        picks, centered = draws[0].to(means.device), draws[1].to(means.device)
    else:
        picks = torch.multinomial(cum_probs, num_samples=num_samples, replacement=True, generator=generator)
        centered = None
    random_indices = vis_indices[picks] if vis_indices is not None else picks
    if centered is None:
        centered = torch.randn((len(random_indices), 3), device=means.device, dtype=torch.float, generator=generator)
    scaled = torch.exp(scales[random_indices]) * centered
    q = quats[random_indices] / quats[random_indices].norm(dim=-1, keepdim=True)
    rots = quat_to_rotmat(q)
    points = means[random_indices] + torch.bmm(rots, scaled[..., None]).squeeze()
    return points, random_indices


def project_pix(p: Tensor, fx: float, fy: float, cx: float, cy: float, c2w: Tensor) -> Tensor:
    """utils/camera_utils.py:147-172 with ``return_z_depths``: (u, v, z) of world points in the camera c2w."""
    cam = (p - c2w[..., :3, 3]) @ c2w[..., :3, :3]
    u = cam[:, 0] * fx / cam[:, 2] + cx
    v = cam[:, 1] * fy / cam[:, 2] + cy
    return torch.stack([u, v, cam[:, 2]], dim=-1)


def get_ideal_sdf(samples: Tensor, depth: Tensor, c2w: Tensor, fx: float, cx: float, cy: float, width: int,
                  height: int, mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """dn_model.py:1497-1546: (rendered depth at the sample's pixel - the sample's depth, validity mask over the samples).
    ``c2w`` as nerfstudio stores it (the y / z flip to the OpenCV frame is applied here, as there); fy = fx, as there."""
    c2w = c2w @ torch.diag(torch.tensor([1, -1, -1, 1], device=c2w.device, dtype=c2w.dtype))
    proj = project_pix(samples, fx, fx, cx, cy, c2w)
    uv = torch.floor(proj[:, :2]).long()
    valid = (uv[:, 0] > 0) & (uv[:, 0] < width) & (uv[:, 1] > 0) & (uv[:, 1] < height)
    if mask is not None:
        inside = valid
        valid = inside.detach().clone()
        valid[inside] = mask[uv[inside, 1], uv[inside, 0]][..., 0].to(torch.bool)
    z_points = proj[valid][..., -1]
    z_ideal = depth[uv[valid, 1], uv[valid, 0], 0]
    return z_ideal - z_points, valid


def scale_rot_to_inv_cov3d_sqrt(scale: Tensor, quat: Tensor) -> Tensor:
    """dn_model.py:2141-2150 with return_sqrt=True: R diag(1 / max(scale, 1e-3))."""
    return quat_to_rotmat(quat) * (1.0 / scale.clamp(min=1e-3))[..., None, :]


def get_density(samples: Tensor, closest: Tensor, means: Tensor, scales: Tensor, quats: Tensor,
                opacities: Tensor) -> Tensor:
    """dn_model.py:1575-1636 (``closest`` [n_samples, k]: the tracked neighbours of every sample's Gaussian)."""
    centers = means[closest]
    inv = scale_rot_to_inv_cov3d_sqrt(torch.exp(scales[closest]), quats[closest])
    op = torch.sigmoid(opacities[closest])
    dist = samples[:, None, :] - centers
    man = inv.transpose(-1, -2) @ dist[..., None]
    maha = (man[..., 0] * man[..., 0]).sum(dim=-1).clamp(min=0.0, max=1e8)
    dens = (op[..., 0] * torch.exp(-1.0 / 2 * maha)).sum(dim=-1)
    over = dens >= 1.0
    dens = torch.where(over, dens / (dens.detach() + 1e-5), dens)
    return dens.clamp(min=1e-4)


def get_sdf(samples: Tensor, closest: Tensor, means: Tensor, scales: Tensor, quats: Tensor, opacities: Tensor) -> Tensor:
    """dn_model.py:1638-1660."""
    return 1 * torch.sqrt(-2.0 * torch.log(get_density(samples, closest, means, scales, quats, opacities)))


@torch.no_grad()
def get_sdf_loss_weight(valid: Tensor, means: Tensor, scales: Tensor, quats: Tensor, cam_pos: Tensor) -> Tensor:
    """dn_model.py:1668-1700, mode "std" — ``valid`` (a mask over the SAMPLES) indexes the Gaussians, as there."""
    if valid.shape[0] != means.shape[0]:
        raise IndexError(f"The shape of the mask {tuple(valid.shape)} at index 0 does not match the shape of the indexed "
                         f"tensor {tuple(means.shape)} at index 0")  # (what torch raises in the reference)
    viewdirs = -means[valid].detach() + cam_pos
    viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
    q = quats[valid] / quats[valid].norm(dim=-1, keepdim=True)
    inv_rots = quat_to_rotmat(invert_quaternion(q))
    return (torch.exp(scales[valid]) * torch.bmm(inv_rots, viewdirs[..., None])[..., 0]).norm(dim=-1)


def sdf_term_due(cfg, step: int) -> bool:
    """dn_model.py:840-844: is the term evaluated at this step?"""
    return bool(getattr(cfg, "use_sdf_loss", False)) and step > int(cfg.apply_sdf_loss_after_iters) and \
        step % int(cfg.apply_sdf_loss_iters) == 0


def sdf_loss(means: Tensor, scales: Tensor, quats: Tensor, opacities: Tensor, knn: Tensor, vis_indices: Tensor,
             add_mask: Optional[Tensor], depth: Tensor, c2w: Tensor, fx: float, cx: float, cy: float, width: int,
             height: int, mask: Optional[Tensor], num_sdf_samples: int,
             draws: Optional[Tuple[Tensor, Tensor]] = None, generator: Optional[torch.Generator] = None) -> Tensor:
    """dn_model.py:838-882 (the value that ``sdf_loss_lambda`` multiplies).  ``knn`` [N, k]: every Gaussian's tracked
    neighbours (``recompute_knn``, :172-195); ``depth`` [H, W, 1]: the rendered depth image; ``c2w`` [3 or 4, 4]."""
    n = means.shape[0]
    num_samples = n if n > num_sdf_samples else num_sdf_samples
    vis = vis_indices
    if add_mask is not None:  # the touch anchors are not sampled
        vis = vis[(~add_mask)[vis].bool()]
    samples, indices = sample_points_in_gaussians(means, scales, quats, num_samples, vis, draws, generator)
    with torch.no_grad():
        closest = knn[indices]
    current = get_sdf(samples, closest, means, scales, quats, opacities)
    if c2w.shape[0] == 3:
        c2w = torch.cat([c2w, torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=c2w.device, dtype=c2w.dtype)], dim=0)
    ideal, valid = get_ideal_sdf(samples.clone().detach(), depth.clone().detach(), c2w, fx, cx, cy, width, height, mask)
    ideal = torch.abs(ideal)
    current = current[valid]
    weight = get_sdf_loss_weight(valid, means, scales, quats, c2w[:3, 3].detach())
    return (torch.abs(ideal - current) / (weight + 1e-5)).mean()
