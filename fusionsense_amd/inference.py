"""Inference callers of the hot path: forward-only consumers of ``get_outputs``.

  * ``get_outputs_for_camera``               SplatfactoModel.get_outputs_for_camera (set_crop + get_outputs)
  * ``get_image_metrics_and_images`` / ``eval_loop``   the evaluation pass with ``num_rays_per_sec`` / ``fps``
    (/root/reference/dn_splatter/dn_model.py:1032-1148, dn_splatter/dn_pipeline.py:233-253, 585-601) and the metric
    classes of dn_splatter/metrics.py (DepthMetrics :113-150, NormalMetrics :153-186, mean_angular_error :58-73)
  * ``backproject_views``                     the body of the mesh exporters that walk the training cameras, render
    depth + normal maps and back-project them into a coloured, oriented point cloud
    (dn_splatter/export_mesh.py:338-455 DepthAndNormalMapsPoisson; the same loop feeds TSDF fusion :836 and the
    level-set extractor :543) with the helpers of dn_splatter/utils/camera_utils.py:70-210 and export_mesh.py:51-91.

The render itself is the HIP path (SplatTrainer.forward under no_grad); what surrounds it here is host-side tensor
glue over [H,W] images, pinned to the reference's own functions by tests/golden/reference_inference.npz.  Poisson /
marching-cubes meshing (open3d) and LPIPS (a pretrained network) are outside the path and not provided.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from .scenes import Camera


@torch.no_grad()
def get_outputs_for_camera(trainer, camera: Camera, obb_box=None) -> Dict[str, Tensor]:
    """``model.get_outputs_for_camera(camera, obb_box)``: eval-mode get_outputs, optional oriented crop box
    (dn_model.py:505-532)."""
    return trainer.forward(camera, crop_box=obb_box)


# ---- dn_splatter/metrics.py -------------------------------------------------------------------------------------
@torch.no_grad()
def depth_metrics(pred: Tensor, gt: Tensor, tolerance: float = 0.1):
    """DepthMetrics.forward: (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3) over gt > tolerance."""
    mask = gt > tolerance
    g, p = gt[mask], pred[mask]
    thresh = torch.max(g / p, p / g)
    a1 = (thresh < 1.25).float().mean()
    a2 = (thresh < 1.25 ** 2).float().mean()
    a3 = (thresh < 1.25 ** 3).float().mean()
    rmse = torch.sqrt(((g - p) ** 2).mean())
    rmse_log = torch.sqrt((torch.log(g) - torch.log(p)) ** 2).nanmean()  # (sqrt BEFORE the mean, as written there)
    abs_rel = (torch.abs(gt - pred)[mask] / g).mean()
    sq_rel = ((gt - pred)[mask] ** 2 / g).mean()
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def mean_angular_error(pred: Tensor, gt: Tensor) -> Tensor:
    """[B,C,H,W] -> [B,H,W] angle between the vectors (radians), dot product clamped to [-1,1]."""
    return torch.acos(torch.clamp(torch.sum(gt * pred, dim=1), -1.0, 1.0))


@torch.no_grad()
def normal_metrics(pred: Tensor, gt: Tensor):
    """NormalMetrics.forward on [B,C,H,W]: (mae, rmse, mean_err, med_err)."""
    b, c, _, _ = gt.shape
    mae = mean_angular_error(pred, gt).mean()
    rmse = torch.sqrt(torch.mean(torch.square(gt - pred), dim=[1, 2, 3])).mean()
    mean_err = torch.mean(torch.abs(gt - pred), dim=[1, 2, 3]).mean()
    med_err = torch.median(torch.abs(gt.reshape(b, c, -1) - pred.reshape(b, c, -1))).mean()
    return mae, rmse, mean_err, med_err


def psnr(gt: Tensor, pred: Tensor, data_range: float = 1.0) -> Tensor:
    """torchmetrics PeakSignalNoiseRatio(data_range=1.0): 10 log10(range^2 / mse)."""
    return 10.0 * torch.log10(data_range ** 2 / torch.mean((gt - pred) ** 2))


def ssim_metric(gt_hw3: Tensor, pred_hw3: Tensor) -> Tensor:
    """SSIM11 of two [H,W,3] images (dn_model.py:244's torchmetrics SSIM): the loss kernel's own sums on the GPU."""
    if pred_hw3.is_cuda:
        from ._lib import load, ptr, stream_ptr
        from .ops import _run
        lib = load()
        H, W = pred_hw3.shape[0], pred_hw3.shape[1]
        dev = pred_hw3.device
        maps = torch.empty(3, H, W, 3, dtype=torch.float32, device=dev)
        sums = torch.empty(lib.fsgs_ssim_l1_num_partials(H, W), 2, dtype=torch.float32, device=dev)
        _run(lib.fsgs_ssim_l1_fwd, (H, W, ptr(pred_hw3.contiguous()), ptr(gt_hw3.contiguous()), maps[0].data_ptr(),
                                    maps[1].data_ptr(), maps[2].data_ptr(), ptr(sums), stream_ptr(dev)), "fsgs_ssim_l1_fwd")
        return sums[:, 1].double().sum().float() / (3.0 * (H - 10) * (W - 10))
    from .losses import ssim
    return ssim(pred_hw3.permute(2, 0, 1), gt_hw3.permute(2, 0, 1))


@torch.no_grad()
def get_image_metrics_and_images(outputs: Dict[str, Tensor], batch: Dict[str, Tensor]):
    """dn_model.py:1032-1148: rgb PSNR / SSIM (masked like the loss), depth metrics against the sensor depth, normal
    metrics against the mono normals; the side-by-side images.  ``rgb_lpips`` is NaN (needs a pretrained network)."""
    dev = outputs["rgb"].device
    gt_rgb = batch["image"].to(dev).float()
    pred_rgb, pred_depth, pred_normal = outputs["rgb"], outputs["depth"], outputs["normal"]
    images = {"img": torch.cat([gt_rgb, pred_rgb], dim=1), "depth": pred_depth, "normal": pred_normal}
    mask = batch["mask"].to(dev).float() if "mask" in batch else None
    g, p = (gt_rgb * mask, pred_rgb * mask) if mask is not None else (gt_rgb, pred_rgb)
    metrics = {"rgb_psnr": float(psnr(g, p)), "rgb_ssim": float(ssim_metric(g, p)), "rgb_lpips": float("nan")}
    if "sensor_depth" in batch:
        gt_depth = batch["sensor_depth"].to(dev).to(torch.float32)
        pd = pred_depth
        if mask is not None:
            gt_depth, pd = gt_depth * mask, pd * mask
        names = ("depth_abs_rel", "depth_sq_rel", "depth_rmse", "depth_rmse_log", "depth_a1", "depth_a2", "depth_a3")
        metrics.update({k: float(v) for k, v in zip(names, depth_metrics(pd.permute(2, 0, 1), gt_depth.permute(2, 0, 1)))})
        images["depth"] = torch.cat([gt_depth, pd], dim=1)
    if "normal" in batch:
        gt_normal = batch["normal"].to(dev).float()
        vals = normal_metrics(pred_normal.permute(2, 0, 1).unsqueeze(0), gt_normal.permute(2, 0, 1).unsqueeze(0))
        metrics.update({k: float(v) for k, v in zip(("normal_mae", "normal_rsme", "normal_mean_err", "normal_med_err"), vals)})
        images["normal"] = torch.cat([gt_normal, pred_normal], dim=1)
    return metrics, images


@torch.no_grad()
def eval_loop(trainer, cameras: List[Camera], batches: List[Dict[str, Tensor]], get_std: bool = False) -> Dict[str, float]:
    """``get_average_eval_image_metrics`` (dn_pipeline.py:233-253, 585-601): every eval camera is rendered with
    get_outputs_for_camera and scored; ``num_rays_per_sec`` = pixels / (render + metrics wall time), ``fps`` =
    that / pixels — the definitions BASELINE's rendered-Mpix/s figure comes from."""
    rows = []
    for cam, batch in zip(cameras, batches):
        t0 = time.time()
        out = get_outputs_for_camera(trainer, cam)
        m, _ = get_image_metrics_and_images(out, batch)  # (float() conversions synchronise, as .item() does there)
        n_rays = cam.height * cam.width
        m["num_rays_per_sec"] = n_rays / (time.time() - t0)
        m["fps"] = m["num_rays_per_sec"] / n_rays
        rows.append(m)
    avg: Dict[str, float] = {}
    for key in rows[0]:
        vals = torch.tensor([r[key] for r in rows])
        if get_std:
            s, mu = torch.std_mean(vals)
            avg[key], avg[f"{key}_std"] = float(mu), float(s)
        else:
            avg[key] = float(torch.mean(vals))
    return avg


# ---- dn_splatter/utils/camera_utils.py + export_mesh.py helpers ----------------------------------------------------
def get_camera_coords(img_size: Tuple[int, int], pixel_offset: float = 0.5, device=None) -> Tensor:
    """[H*W,2] pixel centres (x, y), row-major over the image (img_size = (W, H))."""
    xs, ys = torch.meshgrid(torch.arange(img_size[0], device=device), torch.arange(img_size[1], device=device), indexing="xy")
    return (torch.stack((xs, ys), dim=-1) + pixel_offset).view(-1, 2).float()


def get_means3d_backproj(depths: Tensor, fx: float, fy: float, cx: float, cy: float, img_size: Tuple[int, int],
                         c2w: Tensor, mask: Optional[Tensor] = None):
    """camera_utils.py:92-144: pixel + z-depth -> camera-space point -> world (``c2w`` in OpenCV axes).  Written as the
    reference has it (``p @ inv(R) + t``)."""
    dev = depths.device
    depths = depths.reshape(-1, 1).float()
    c2w = c2w.float().to(dev)
    coords = get_camera_coords(img_size, device=dev)
    means3d = torch.empty(coords.shape[0], 3, dtype=torch.float32, device=dev)
    means3d[:, 0] = (coords[:, 0] - cx) * depths[:, 0] / fx
    means3d[:, 1] = (coords[:, 1] - cy) * depths[:, 0] / fy
    means3d[:, 2] = depths[:, 0]
    if mask is not None:
        means3d, coords = means3d[mask], coords[mask]
    # (tiny 3x3 inverse on the host: no solver library on the device path)
    rinv = torch.linalg.inv(c2w[..., :3, :3].cpu()).to(dev)
    return means3d @ rinv + c2w[..., :3, 3], coords


def project_pix(p: Tensor, fx: float, fy: float, cx: float, cy: float, c2w: Tensor, return_z_depths: bool = False) -> Tensor:
    """camera_utils.py:147-172."""
    c2w = c2w.to(p.device)
    pc = (p - c2w[..., :3, 3]) @ c2w[..., :3, :3]
    u = pc[:, 0] * fx / pc[:, 2] + cx
    v = pc[:, 1] * fy / pc[:, 2] + cy
    return torch.stack([u, v, pc[:, 2]], dim=-1) if return_z_depths else torch.stack([u, v], dim=-1)


def get_colored_points_from_depth(depths: Tensor, rgbs: Tensor, c2w: Tensor, fx: float, fy: float, cx: float, cy: float,
                                  img_size: Tuple[int, int], mask: Optional[Tensor] = None):
    """camera_utils.py:175-210."""
    points, _ = get_means3d_backproj(depths.float(), fx, fy, cx, cy, img_size, c2w.float())
    colors = rgbs.reshape(-1, 3)
    if mask is not None:
        mask = torch.as_tensor(mask, device=depths.device)
        return points[mask], colors[mask]
    return points, colors


def pick_indices_at_random(valid_mask: Tensor, samples_per_frame: int, generator=None) -> Tensor:
    """export_mesh.py:51-56."""
    indices = torch.nonzero(torch.ravel(valid_mask))
    if samples_per_frame < len(indices):
        which = torch.randperm(len(indices), generator=generator)[:samples_per_frame].to(indices.device)
        indices = indices[which]
    return torch.ravel(indices)


def find_depth_edges(depth_im: Tensor, threshold: float = 0.01, dilation_itr: int = 3) -> Tensor:
    """export_mesh.py:59-91: Laplacian of the inverse depth above ``threshold``, dilated ``dilation_itr`` times with
    a 3x3 box; [H,W,1] of {0,1}.  The two tiny stencils are evaluated with shifted adds (no convolution library)."""
    inv = 1.0 / (depth_im[..., 0] + 1e-6)
    p = F.pad(inv, (1, 1, 1, 1))
    lap = p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] - 4.0 * inv
    d = (lap > threshold) * 1.0
    for _ in range(dilation_itr):
        q = F.pad(d, (1, 1, 1, 1))
        d = sum(q[i:i + d.shape[0], j:j + d.shape[1]] for i in range(3) for j in range(3))
    return ((d > 0.0) * 1.0)[..., None]


@torch.no_grad()
def backproject_views(trainer, cameras: List[Camera], batches: Optional[List[Dict[str, Tensor]]] = None,
                      total_points: int = 2_000_000, use_masks: bool = True, filter_edges_from_depth_maps: bool = False,
                      edge_threshold: float = 0.004, edge_dilation_iterations: int = 10, generator=None):
    """DepthAndNormalMapsPoisson up to its point cloud (export_mesh.py:338-455, normal_method = "normal_maps"): every
    training camera is rendered (get_outputs_for_camera), ``samples_per_frame`` pixels are drawn, their depths
    back-projected with the OpenCV-axes camera-to-world, colours taken from the rgb render, normals from the normal
    render ([0,1] -> [-1,1], flipped to OpenGL axes, normalised, rotated to the world).  Returns (points, colors,
    normals) — what the reference hands to open3d's Poisson reconstruction."""
    n = len(cameras)
    samples_per_frame = (total_points + n) // n
    points, colors, normals = [], [], []
    for i, cam in enumerate(cameras):
        out = get_outputs_for_camera(trainer, cam)
        depth_map = out["depth"].clone()
        dev = depth_map.device
        c2w = torch.eye(4, device=dev)
        c2w[:3, :4] = cam.c2w.to(dev)
        c2w = (c2w @ torch.diag(torch.tensor([1.0, -1.0, -1.0, 1.0], device=dev)))[:3, :4]
        H, W = cam.height, cam.width
        valid = (find_depth_edges(depth_map, edge_threshold, edge_dilation_iterations) < 0.2
                 if filter_edges_from_depth_maps else depth_map)
        idx = pick_indices_at_random(valid, samples_per_frame, generator)
        if len(idx) == 0:
            continue
        mask = batches[i].get("mask") if batches is not None else None
        if mask is not None and use_masks:
            depth_map[~mask.to(dev).bool()] = 0
        xyz, rgb = get_colored_points_from_depth(depth_map, out["rgb"], c2w, cam.fx, cam.fy, cam.cx, cam.cy, (W, H), idx)
        nm = (2 * out["normal"].reshape(-1, 3) - 1) @ torch.diag(torch.tensor([1.0, -1.0, -1.0], device=dev))
        nm = F.normalize(nm.T, p=2, dim=0)
        nm = (c2w[:3, :3] @ nm).T[idx]
        points.append(xyz); colors.append(rgb); normals.append(nm)
    return torch.cat(points), torch.cat(colors), torch.cat(normals)


# ---- level-set extraction (dn_model.py:1706-1946; the LevelSetExtractor of export_mesh.py:486-600) ---------------
def knn_drop_first(x: Tensor, y: Tensor, k: int) -> Tensor:
    """``dn_splatter.utils.knn.knn_sk(x, y, k)`` (utils/knn.py:29-44): for every row of ``y`` the indices of rows of
    ``x`` ranked 2nd .. (k+1)-th nearest — sklearn is asked for k + 1 neighbours and the first column is dropped (the
    self-match when y == x; for other queries it is the true nearest neighbour that goes, which every caller on this
    path inherits).  Exact fp32 differences (no |a|^2 + |b|^2 - 2ab), brute force in query chunks on ``x``'s device."""
    x, y = x.float(), y.float().to(x.device)
    n, m = x.shape[0], y.shape[0]
    kk = min(k + 1, n)
    if x.is_cuda and kk <= 33:
        # on the device: one launch of the brute-force k-NN kernel (fsgs_knn_points; ~100 ms for a full 800 x 800 frame
        # against 300 k Gaussians, where the chunked torch formulation below needs hundreds of [chunk, N] passes)
        from ._lib import load, ptr, stream_ptr
        from .ops import _run
        out = torch.empty(m, kk - 1, dtype=torch.int64, device=x.device)
        xc, yc = x.contiguous(), y.contiguous()
        _run(load().fsgs_knn_points, (m, ptr(yc), n, ptr(xc), kk, 1, ptr(out), stream_ptr(x.device)), "fsgs_knn_points")
        return out
    out = torch.empty(m, kk - 1, dtype=torch.int64, device=x.device)
    chunk = max(1, min(m, (1 << 28) // max(n, 1)))
    for s in range(0, m, chunk):
        q = y[s:s + chunk]
        d = (q[:, None, 0] - x[None, :, 0]) ** 2
        d += (q[:, None, 1] - x[None, :, 1]) ** 2
        d += (q[:, None, 2] - x[None, :, 2]) ** 2
        out[s:s + chunk] = d.topk(kk, dim=1, largest=False).indices[:, 1:]
    return out


def _quat_to_rotmat(q: Tensor) -> Tensor:
    from .legacy import quat_to_rotmat
    return quat_to_rotmat(q)


def gaussian_density(samples: Tensor, idx: Tensor, means: Tensor, inv_scaled_rot: Tensor, strengths: Tensor,
                     want_grad: bool = False):
    """The density field of dn_model.py:1812-1844 (also get_density :1580-1635) at ``samples`` [S,3] over each
    sample's ``idx`` [S,k] Gaussians: sum_k sigmoid(opacity_k) exp(-1/2 |M_k^T (s - mu_k)|^2) with
    M = R(q) diag(1 / max(scale, 1e-3)) (scale_rot_to_inv_cov3d(return_sqrt=True), :2141-2150); values >= 1 are
    renormalised d / (d + 1e-5).  ``want_grad``: also sum_k w_k (M_k M_k^T (s - mu_k)) — the negative density gradient
    up to the factor the "analytical" normal normalises away (:1900-1919)."""
    shift = samples[:, None, :] - means[idx]                       # [S,k,3]
    M = inv_scaled_rot[idx]                                        # [S,k,3,3]
    # M^T shift and M man as explicit 3-term sums: tens of millions of 3x3 products per pass are elementwise work, not
    # a batched GEMM (a [32 M]-batch bmm of 3x3 matrices faulted in the BLAS library on the GPU box)
    man = (M * shift[..., :, None]).sum(dim=-2)                    # [S,k,3]: sum_i M[i][j] shift[i]
    d2 = (man * man).sum(dim=-1).clamp(min=0.0, max=1e8)
    w = strengths[idx][..., 0] * torch.exp(-0.5 * d2)              # [S,k]
    dens = w.sum(dim=-1)
    big = dens >= 1.0
    dens = torch.where(big, dens / (dens + 1e-5), dens)
    if want_grad:
        return dens, (w[..., None] * (M * man[..., None, :]).sum(dim=-1)).sum(dim=-2)
    return dens


@torch.no_grad()
def level_surface_points_from_render(params: Dict[str, Tensor], normals_world: Tensor, depth: Tensor, rgb: Tensor,
                                     camera: Camera, num_samples: int, mask: Optional[Tensor] = None,
                                     surface_levels=(0.1, 0.3, 0.5), return_normal: str = "closest_gaussian",
                                     knn_to_track: int = 16, sample_fn=None, n_points_per_pass: int = 2_000_000):
    """``DNSplatterModel.compute_level_surface_points`` (dn_model.py:1706-1946) after its ``get_outputs`` call: every
    pixel with depth is back-projected, 21 samples are laid along its ray over +-3 standard deviations of the point's
    (2nd-)closest Gaussian in that direction, the density of the point's ``knn_to_track`` Gaussians is evaluated at them,
    and for each level the first crossing from below is located by linear interpolation.  Returns
    {level: {"points", "normals", "colors"}} like the reference; ``sample_fn(n_available, n_wanted) -> indices`` replaces
    its ``random.sample`` (default: exactly that)."""
    import random
    dev = depth.device
    H, W = camera.height, camera.width
    c2w_gl = camera.c2w.to(device=dev, dtype=torch.float32)        # [3,4], OpenGL axes as nerfstudio stores it
    campos = c2w_gl[None, :3, 3]                                   # [1,3] (camera.camera_to_worlds[..., :3, 3])
    c2w = c2w_gl @ torch.diag(torch.tensor([1.0, -1.0, -1.0, 1.0], device=dev))
    means, quats = params["means"].detach().to(dev), params["quats"].detach().to(dev)
    scales, opac = params["scales"].detach().to(dev), params["opacities"].detach().to(dev)
    depth = depth.reshape(H, W, 1)
    points, colors = get_colored_points_from_depth(depths=depth, rgbs=rgb, fx=camera.fx, fy=camera.fy, cx=camera.cx,
                                                   cy=camera.cy, img_size=(W, H), c2w=c2w)
    points, colors = points.view(H, W, -1), colors.reshape(H, W, 3)
    if mask is not None:
        mask = mask.to(dev)
        points = points * mask
        depth = depth * mask
    keep = ~(depth <= 0.0)[..., 0]
    points, colors = points[keep], colors[keep]
    pixel_ids = torch.nonzero(keep.reshape(-1)).reshape(-1)
    closest = knn_drop_first(means, points, knn_to_track)

    viewdirs = -means + campos
    viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
    qn = quats / quats.norm(dim=-1, keepdim=True)
    inv_rots = _quat_to_rotmat(qn * torch.tensor([1.0, -1.0, -1.0, -1.0], device=dev))  # invert_quaternion (:2153-2163)
    stds = (torch.exp(scales) * (inv_rots * viewdirs[:, None, :]).sum(dim=-1)).norm(dim=-1)  # (R^-1 viewdir, no BLAS)
    points_stds = stds[closest][..., 0]

    n_in_range = 21
    points_range = torch.linspace(-3, 3, n_in_range, device=dev).view(1, -1, 1) * points_stds[..., None, None]
    cam_to_samples = F.normalize(points - campos, dim=-1)
    samples = (points[:, None, :] + points_range * cam_to_samples[:, None, :]).view(-1, 3)
    samples_idx = closest[:, None, :].expand(-1, n_in_range, -1).reshape(-1, closest.shape[1])
    strengths = torch.sigmoid(opac)
    inv_sr = _quat_to_rotmat(quats) * (1.0 / torch.exp(scales).clamp(min=1e-3))[..., None, :]
    densities = torch.empty(samples.shape[0], dtype=torch.float32, device=dev)
    for s in range(0, samples.shape[0], n_points_per_pass):
        e = min(samples.shape[0], s + n_points_per_pass)
        densities[s:e] = gaussian_density(samples[s:e], samples_idx[s:e], means, inv_sr, strengths)
    densities = densities.reshape(-1, n_in_range)

    out = {}
    for level in surface_levels:
        under, above = densities - level < 0, densities - level > 0
        first = above.to(torch.uint8).max(dim=-1, keepdim=True).indices  # first sample above the level (0: none / the first)
        empty = ~under[..., 0] | (first[..., 0] == 0)
        vd, vr, vf = densities[~empty], points_range[~empty][..., 0], first[~empty]
        v1, v0 = vd.gather(-1, vf).view(-1), vd.gather(-1, vf - 1).view(-1)
        t1, t0 = vr.gather(-1, vf).view(-1), vr.gather(-1, vf - 1).view(-1)
        t = (level - v0) / (v1 - v0) * (t1 - t0) + t0
        ipts = points[~empty] + t[:, None] * cam_to_samples[~empty]
        icol = colors[~empty]
        idx_k = closest[~empty]
        if return_normal == "analytical":
            _, g = gaussian_density(ipts, idx_k, means, inv_sr, strengths, want_grad=True)
            inrm = -F.normalize(g, dim=-1)
        elif return_normal == "closest_gaussian":
            inrm = normals_world.to(dev)[idx_k[..., 0]]
        else:
            raise NotImplementedError(return_normal)
        n_av = ipts.shape[0]
        want = num_samples if num_samples < n_av else n_av
        pick = sample_fn(n_av, want) if sample_fn is not None else random.sample(range(n_av), want)
        pick = torch.as_tensor(pick, device=dev, dtype=torch.int64)
        out[level] = {"points": ipts[pick], "normals": inrm[pick], "colors": icol[pick],
                      "pixel_ids": pixel_ids[~empty][pick]}  # (extra: the flat pixel index every point came from)
    return out


@torch.no_grad()
def compute_level_surface_points(trainer, camera: Camera, num_samples: int, mask: Optional[Tensor] = None,
                                 surface_levels=(0.1, 0.3, 0.5), return_normal: str = "closest_gaussian",
                                 knn_to_track: int = 16, sample_fn=None):
    """``model.compute_level_surface_points(camera, num_samples, mask, surface_levels, return_normal)`` over the HIP
    forward: get_outputs (eval mode) renders depth + rgb and leaves the world-space normals of the Gaussians
    (``gauss_params["normals"]``, dn_model.py:634), the rest is :func:`level_surface_points_from_render`."""
    o = trainer.forward(camera)
    return level_surface_points_from_render(trainer.params, o["normals_world"], o["depth"], o["rgb"], camera, num_samples,
                                            mask, surface_levels, return_normal, knn_to_track, sample_fn)


@torch.no_grad()
def extract_level_sets(trainer, cameras: List[Camera], batches: Optional[List[Dict[str, Tensor]]] = None,
                       total_points: int = 100_000, use_masks: bool = False, surface_levels=(0.1, 0.3, 0.5),
                       return_normal: str = "closest_gaussian", sample_fn=None):
    """LevelSetExtractor.main up to its point clouds (export_mesh.py:506-583): ``samples_per_frame`` level-surface
    points from every training camera, concatenated per level -> {level: (points, colors, normals)} — what the
    reference hands to open3d's Poisson reconstruction (off the path, not provided)."""
    n = len(cameras)
    per_frame = (total_points + n) // n
    acc = {lv: {"points": [], "colors": [], "normals": []} for lv in surface_levels}
    for i, cam in enumerate(cameras):
        mask = batches[i].get("mask") if (batches is not None and use_masks) else None
        fo = compute_level_surface_points(trainer, cam, per_frame, mask, surface_levels, return_normal, sample_fn=sample_fn)
        for lv in surface_levels:
            for key in acc[lv]:
                acc[lv][key].append(fo[lv][key])
    return {lv: tuple(torch.cat(acc[lv][k]) for k in ("points", "colors", "normals")) for lv in surface_levels}
