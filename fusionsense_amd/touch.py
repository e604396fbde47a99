"""FusionSense-specific pruning / seeding around the rasterizer: touch-patch
anchors and visual-hull pruning.  Mirrors of the pure-tensor helpers and callbacks in
/root/reference/dn_splatter/dn_model.py (cited per function); masks are a few elementwise
torch ops over [N], the row movement itself goes through DensifyStrategy (libfsgs.so).
Pinned against the reference's own functions by tests/golden/reference_helpers.npz."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import Tensor

from .scenes import rgb_to_sh


def points_in_non_aabb(point_cloud: Tensor, box_vertices: Tensor) -> Tensor:
    """dn_model.py:1996-2034: points inside an oriented box given by its 8 vertices (edges
    0->1, 0->2, 0->4).  Inclusive bounds."""
    box_center = box_vertices.mean(dim=0)
    axes = torch.stack([box_vertices[1] - box_vertices[0], box_vertices[2] - box_vertices[0],
                        box_vertices[4] - box_vertices[0]])
    axes = torch.nn.functional.normalize(axes, dim=1)
    rel_p = (point_cloud - box_center) @ axes.T  # [N,3]
    rel_c = (box_vertices - box_center) @ axes.T  # [8,3]
    lo = rel_c.min(dim=0).values
    hi = rel_c.max(dim=0).values
    return ((rel_p >= lo) & (rel_p <= hi)).all(dim=-1)


def rotate_vector_to_vector(v1: Tensor, v2: Tensor) -> Tensor:
    """dn_model.py:2058-2089: rotation matrices taking v1 onto v2 (Rodrigues)."""
    if v1.dim() == 1:
        v1, v2 = v1[None], v2[None]
    N = v1.shape[0]
    u = v1 / torch.norm(v1, dim=-1, keepdim=True)
    Ru = v2 / torch.norm(v2, dim=-1, keepdim=True)
    eye = torch.eye(3, device=v1.device, dtype=v1.dtype).unsqueeze(0).repeat(N, 1, 1)
    c = (u * Ru).sum(-1, keepdim=True)  # [N,1]
    K = torch.bmm(Ru.unsqueeze(2), u.unsqueeze(1)) - torch.bmm(u.unsqueeze(2), Ru.unsqueeze(1))
    ans = eye + K + (K @ K) / (1 + c)[..., None]
    same = (torch.abs(c - 1.0) < 1e-10).squeeze(-1)
    opp = (torch.abs(c + 1.0) < 1e-10).squeeze(-1)
    ans[same] = torch.eye(3, device=v1.device, dtype=v1.dtype)
    ans[opp] = -torch.eye(3, device=v1.device, dtype=v1.dtype)
    return ans


def matrix_to_quaternion(R: Tensor) -> Tensor:
    """dn_model.py:2092-2138 (wxyz), vectorised: the reference loops over matrices in Python."""
    if R.dim() == 2:
        R = R[None]
    m00, m01, m02 = R[:, 0, 0], R[:, 0, 1], R[:, 0, 2]
    m10, m11, m12 = R[:, 1, 0], R[:, 1, 1], R[:, 1, 2]
    m20, m21, m22 = R[:, 2, 0], R[:, 2, 1], R[:, 2, 2]
    tr = m00 + m11 + m22
    b0 = tr > 0
    b1 = ~b0 & (m00 > m11) & (m00 > m22)
    b2 = ~b0 & ~b1 & (m11 > m22)

    def safe_sqrt(x):
        return torch.sqrt(torch.clamp(x, min=1e-30))

    S0 = safe_sqrt(tr + 1.0) * 2
    q0 = torch.stack([0.25 * S0, (m21 - m12) / S0, (m02 - m20) / S0, (m10 - m01) / S0], -1)
    S1 = safe_sqrt(1.0 + m00 - m11 - m22) * 2
    q1 = torch.stack([(m21 - m12) / S1, 0.25 * S1, (m01 + m10) / S1, (m02 + m20) / S1], -1)
    S2 = safe_sqrt(1.0 + m11 - m00 - m22) * 2
    q2 = torch.stack([(m02 - m20) / S2, (m01 + m10) / S2, 0.25 * S2, (m12 + m21) / S2], -1)
    S3 = safe_sqrt(1.0 + m22 - m00 - m11) * 2
    q3 = torch.stack([(m10 - m01) / S3, (m02 + m20) / S3, (m12 + m21) / S3, 0.25 * S3], -1)
    return torch.where(b0[:, None], q0, torch.where(b1[:, None], q1, torch.where(b2[:, None], q2, q3)))


def invert_quaternion(quat: Tensor) -> Tensor:
    """dn_model.py:2153-2163 (wxyz conjugate)."""
    return quat * torch.tensor([1, -1, -1, -1], device=quat.device, dtype=quat.dtype)


def box_frames(touch_patches: List[Dict[str, Tensor]], device=None) -> Tensor:
    """[B, 18] float32: centre, unit axes (rows), lower and upper bounds of every non-empty patch's oriented box — the
    per-box half of points_in_non_aabb (dn_model.py:2008-2027), computed once per run (DensifyStrategy.stage)."""
    rows = []
    for patch in touch_patches:
        if patch["points_xyz"].shape[0] == 0:
            continue
        bv = patch["bbox"].to(torch.float32)
        center = bv.mean(dim=0)
        axes = torch.nn.functional.normalize(torch.stack([bv[1] - bv[0], bv[2] - bv[0], bv[4] - bv[0]]), dim=1)
        rel_c = (bv - center) @ axes.T
        rows.append(torch.cat([center, axes.reshape(-1), rel_c.min(dim=0).values, rel_c.max(dim=0).values]))
    out = torch.stack(rows) if rows else torch.zeros(0, 18)
    return out.to(device) if device is not None else out


def touch_aabb_mask(means: Tensor, touch_patches: List[Dict[str, Tensor]], frames: Optional[Tensor] = None) -> Tensor:
    """Union of the oriented-box tests over all non-empty patches (dn_model.py:1173-1184,
    1284-1294).  On the device: one launch over the boxes' frames (``frames`` = box_frames(touch_patches), computed
    here when the caller has not kept them)."""
    if means.is_cuda:
        from .ops import points_in_boxes
        if frames is None:
            frames = box_frames(touch_patches, means.device)
        return points_in_boxes(means, frames)
    mask = torch.zeros(means.shape[0], dtype=torch.bool, device=means.device)
    for patch in touch_patches:
        if patch["points_xyz"].shape[0] > 0:
            mask |= points_in_non_aabb(means, patch["bbox"].to(means.device))
    return mask


def hull_prune_mask(means: Tensor, visual_hull: Tensor, scale_factor: float,
                    add_mask: Optional[Tensor]) -> Tensor:
    """dn_model.py:1249-1276: Gaussians near the object (<= 0.2 s from the hull centre) whose
    distance to the hull point set is in (0.005 s, 0.02 s] are culled; touch anchors never."""
    center = visual_hull.mean(dim=0)
    close = torch.norm(means - center, dim=1) <= 0.2 * scale_factor
    if not means.is_cuda:
        raise ValueError("hull_prune_mask runs on the GPU (fsgs_nearest_point): there is no CPU path")
    # the close rows' distances (exact differences, no [n, n_hull] matrix), selected ON the device: the kernel skips
    # the other rows (+inf) — no wait for the host to learn how many rows are close, no gather / scatter
    from .ops import nearest_point
    d = nearest_point(means, visual_hull, active=close)
    mask = close & (d > 0.005 * scale_factor) & (d <= 0.02 * scale_factor)
    if add_mask is not None:
        mask = mask & ~add_mask
    return mask


def hull_grid(visual_hull: Tensor, reach: float, max_cells: int = 1 << 22) -> Optional[Dict]:
    """The visual hull's points binned into a uniform grid whose cell edge just exceeds ``reach`` (hull_pruning's outer
    radius, 0.02 * scale_factor: dn_model.py:1266-1270), for fsgs_refine_mark's in-launch search: a Gaussian's nearest hull
    point matters to hull_pruning only if it is within ``reach``, and then it lies in one of the 27 cells around the
    Gaussian.  Built once per hull (the hull is scene metadata).  Returns None when the grid would need more than
    ``max_cells`` cells (a hull that is huge against its own pruning radius: the exhaustive search stays).
    points [H, 4] fp32 sorted by cell, cells [n_cells + 1] int32 (first point of every cell), origin / inv_cell / dims as
    the kernel applies them: cell(p) = floor((p - origin) * inv_cell), id = (z * dims[1] + y) * dims[0] + x."""
    hull = visual_hull.to(torch.float32)
    if hull.shape[0] == 0:
        return None
    cell = torch.tensor(float(reach), dtype=torch.float32, device=hull.device) * 1.001
    inv = 1.0 / cell
    origin = hull.min(dim=0).values - cell
    g = torch.floor((hull - origin) * inv).to(torch.int64)      # (the kernel's own expression, operation for operation)
    dims = [int(x) + 2 for x in g.max(dim=0).values.tolist()]
    n_cells = dims[0] * dims[1] * dims[2]
    if n_cells > max_cells:
        return None
    ids = (g[:, 2] * dims[1] + g[:, 1]) * dims[0] + g[:, 0]
    order = torch.argsort(ids, stable=True)
    points = torch.zeros(hull.shape[0], 4, dtype=torch.float32, device=hull.device)
    points[:, :3] = hull[order]
    cells = torch.zeros(n_cells + 1, dtype=torch.int32, device=hull.device)
    cells[1:] = torch.cumsum(torch.bincount(ids, minlength=n_cells), 0).to(torch.int32)
    return dict(points=points.contiguous(), cells=cells, origin=[float(x) for x in origin.tolist()], inv_cell=float(inv),
                dims=dims)


def make_touch_gaussians(points: Tensor, normals: Tensor, rgb: Tensor, gel_scale_factor: float,
                         sh_degree: int) -> Dict[str, Tensor]:
    """Rows appended by add_touch_patch (dn_model.py:1190-1224): raw opacity 1.0, log-scale
    log(g) with the z axis /3, orientation rotating +z onto the contact normal."""
    n = points.shape[0]
    dev = points.device
    K = (sh_degree + 1) ** 2
    shs = torch.zeros(n, K, 3, device=dev)
    if sh_degree > 0:
        shs[:, 0, :] = rgb_to_sh(rgb)
    else:
        shs[:, 0, :] = torch.logit(rgb, eps=1e-10)
    g = torch.tensor(float(gel_scale_factor), device=dev)
    scales = torch.log(g.repeat(n, 3))
    scales[:, 2] = torch.log(g / 3)
    z = torch.tensor([0.0, 0.0, 1.0], device=dev).repeat(n, 1)
    quats = matrix_to_quaternion(rotate_vector_to_vector(z, normals))
    return dict(means=points, opacities=torch.ones(n, 1, device=dev), features_dc=shs[:, 0, :].contiguous(),
                features_rest=shs[:, 1:, :].contiguous(), scales=scales, quats=quats)
