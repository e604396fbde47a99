"""CPU oracle for the FusionSense training loss: ``DNSplatterModel.get_loss_dict``
(/root/reference/dn_splatter/dn_model.py:673-925) on top of splatfacto's photometric term.
TEST INFRASTRUCTURE ONLY — never imported by the product path (fusionsense_amd/, bench.py's timed region).

Pinning:
  * everything FusionSense's own code adds (mask handling :702-714, EdgeAwareLogL1 with the valid mask :721-736 and
    losses.py:177-214, TV on depth :752-760 and losses.py:269-285, normal L1 + TV :809-815, min-scale :817-819,
    touch-normal MSE :893-904; round 4: normals from depth :774-795, EdgeAwareTV :757-760, the cosine term :807-813,
    the L1 / LogL1 / MSE depth-loss types :731-736) is PINNED: tests/golden/reference_loss.npz holds values and gradients produced by
    executing the reference's own method (tests/golden/make_reference_loss_goldens.py), and
    tests/test_oracle.py::test_loss_oracle_matches_reference_goldens checks this file against them.
  * the photometric term (1-l)*L1 + l*(1-SSIM) comes from nerfstudio 1.1.3 ``SplatfactoModel.get_loss_dict`` with
    torchmetrics' ``StructuralSimilarityIndexMeasure(data_range=1.0, kernel_size=11)`` swapped in at
    dn_model.py:244 — neither package is installed: restated from their published algorithm, PARITY UNPINNED
    (known answers only: SSIM(x, x) = 1, symmetry, the constant-image closed form).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F
from torch import Tensor


@dataclass
class LossConfig:
    """DNSplatterModelConfig (dn_model.py:60-141) with the values scripts/train.py passes on from
    /root/reference/configs/config.py:9-15."""
    ssim_lambda: float = 0.2            # nerfstudio default
    use_depth_loss: bool = True         # configs/config.py:9
    sensor_depth_lambda: float = 0.2    # configs/config.py:11
    mono_depth_lambda: float = 0.2      # dn_model.py:70 (used only when the batch carries "mono_depth")
    depth_tolerance: float = 0.1        # dn_model.py:64
    use_depth_smooth_loss: bool = True  # configs/config.py:12
    smooth_loss_lambda: float = 0.1     # dn_model.py:74
    use_normal_loss: bool = True        # configs/config.py:14
    use_normal_tv_loss: bool = True     # dn_model.py:82
    normal_lambda: float = 0.4          # configs/config.py:10
    two_d_gaussians: bool = True        # dn_model.py:98
    touch_normal_loss_lambda: float = 1.0  # dn_model.py:901
    # switches that are off in FusionSense's configuration (round 4; pinned by the ndepth / eatv / cosine / l1 / logl1 /
    # mse cases of reference_loss.npz, which the reference's own get_loss_dict produced)
    normal_supervision: str = "mono"    # dn_model.py:84 default "depth"; configs/config.py:15 passes "mono"
    smooth_loss_type: str = "TV"        # dn_model.py:66 ("TV" | "EdgeAwareTV")
    use_normal_cosine_loss: bool = False  # dn_model.py:80
    depth_loss_type: str = "EdgeAwareLogL1"  # dn_model.py:62 ("EdgeAwareLogL1" | "L1" | "LogL1" | "MSE" | "HuberL1")


def ssim_torchmetrics(pred: Tensor, gt: Tensor, kernel_size: int = 11, sigma: float = 1.5,
                      data_range: float = 1.0) -> Tensor:
    """torchmetrics 'structural_similarity_index_measure' (gaussian_kernel=True, reduction mean) for one
    [3,H,W] pair: reflect-pad by 5, depthwise 11x11 Gaussian (outer product of the normalised 1-D window),
    SSIM map, crop the pad again, mean.  Recalled; unpinned."""
    C = pred.shape[0]
    dt = pred.dtype
    x = torch.arange(kernel_size, dtype=dt) - (kernel_size - 1) / 2.0
    g1 = torch.exp(-(x / sigma) ** 2 / 2)
    g1 = g1 / g1.sum()
    k2 = (g1[:, None] * g1[None, :]).expand(C, 1, kernel_size, kernel_size)
    pad = (kernel_size - 1) // 2
    p = F.pad(pred[None], (pad, pad, pad, pad), mode="reflect")
    t = F.pad(gt[None], (pad, pad, pad, pad), mode="reflect")
    stack = torch.cat([p, t, p * p, t * t, p * t], 0)  # [5,C,H+2p,W+2p]
    o = F.conv2d(stack, k2, groups=C)
    mu_p, mu_t, e_pp, e_tt, e_pt = o[0], o[1], o[2], o[3], o[4]
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s_pp, s_tt, s_pt = e_pp - mu_p ** 2, e_tt - mu_t ** 2, e_pt - mu_p * mu_t
    m = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p ** 2 + mu_t ** 2 + c1) * (s_pp + s_tt + c2))
    return m[..., pad:-pad, pad:-pad].mean()


def photometric_loss(pred_rgb: Tensor, image: Tensor, mask: Optional[Tensor], ssim_lambda: float) -> Tensor:
    """nerfstudio SplatfactoModel.get_loss_dict: both images are multiplied by the mask, then
    (1-l)*mean|gt-pred| + l*(1-SSIM)."""
    gt, pred = image, pred_rgb
    if mask is not None:
        gt = gt * mask
        pred = pred * mask
    l1 = torch.abs(gt - pred).mean()
    sim = 1 - ssim_torchmetrics(gt.permute(2, 0, 1), pred.permute(2, 0, 1))
    return (1 - ssim_lambda) * l1 + ssim_lambda * sim


def edge_aware_log_l1(pred: Tensor, gt: Tensor, rgb: Tensor, mask: Optional[Tensor]) -> Tensor:
    """losses.py:177-214, scalar form: log(1+|d-d*|) weighted by exp(-mean_c|dI|) along x and y, each averaged
    over the VALID pixels of its shifted grid."""
    logl1 = torch.log(1 + torch.abs(pred - gt))
    gx = torch.mean(torch.abs(rgb[:, :-1, :] - rgb[:, 1:, :]), -1, keepdim=True)
    gy = torch.mean(torch.abs(rgb[:-1, :, :] - rgb[1:, :, :]), -1, keepdim=True)
    lx = torch.exp(-gx) * logl1[:, :-1, :]
    ly = torch.exp(-gy) * logl1[:-1, :, :]
    if mask is not None:
        lx = lx[mask[:, :-1, :]]
        ly = ly[mask[:-1, :, :]]
    return lx.mean() + ly.mean()


def tv(pred: Tensor) -> Tensor:
    """losses.py:269-285."""
    return torch.mean(torch.abs(pred[:, :-1, :] - pred[:, 1:, :])) + torch.mean(torch.abs(pred[:-1, :, :] - pred[1:, :, :]))


def edge_aware_tv(depth: Tensor, rgb: Tensor) -> Tensor:
    """losses.py:241-266 (EdgeAwareTV): |d depth| weighted by exp(-mean_c |d rgb|), mean over each shifted grid."""
    gdx = torch.abs(depth[:, :-1, :] - depth[:, 1:, :])
    gdy = torch.abs(depth[:-1, :, :] - depth[1:, :, :])
    gix = torch.mean(torch.abs(rgb[:, :-1, :] - rgb[:, 1:, :]), -1, keepdim=True)
    giy = torch.mean(torch.abs(rgb[:-1, :, :] - rgb[1:, :, :]), -1, keepdim=True)
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


def normal_from_depth(depth: Tensor, fx: float, fy: float, cx: float, cy: float) -> Tensor:
    """utils/normal_utils.py:23-46 with c2w = identity, smooth = False (as dn_model.py:779-789 calls it) over
    utils/camera_utils.py:92-144: every pixel centre (x + 0.5, y + 0.5) is back-projected with its z-depth, the
    normal of an interior pixel is normalize(cross(right - left, top - bottom)), the one-pixel border is zero."""
    H, W = depth.shape[0], depth.shape[1]
    d = depth.reshape(H, W)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=d.dtype), torch.arange(W, dtype=d.dtype), indexing="ij")
    xyz = torch.stack([(xs + 0.5 - cx) * d / fx, (ys + 0.5 - cy) * d / fy, d], dim=-1)
    l2r = xyz[1:H - 1, 2:W] - xyz[1:H - 1, 0:W - 2]
    b2t = xyz[0:H - 2, 1:W - 1] - xyz[2:H, 1:W - 1]
    n = F.normalize(torch.cross(l2r, b2t, dim=-1), p=2, dim=-1)
    return F.pad(n.permute(2, 0, 1), (1, 1, 1, 1)).permute(1, 2, 0)


def mean_angular_error_as_called(pred_hw3: Tensor, gt_hw3: Tensor) -> Tensor:
    """dn_model.py:807-813 calls metrics.mean_angular_error (metrics.py:58-73, written for [B,3,H,W]) with [3,H,W]
    tensors: its ``sum(dim=1)`` then runs over the image ROWS — one dot product per (channel, column).  Reproduced as
    called: acos(clamp(sum_y gt * pred, -1, 1)) over [3, W], mean."""
    p = (pred_hw3.permute(2, 0, 1) - 1) / 2
    g = (gt_hw3.permute(2, 0, 1) - 1) / 2
    return torch.acos(torch.clamp(torch.sum(g * p, dim=1), -1.0, 1.0)).mean()


def plain_depth_loss(kind: str, pred: Tensor, gt: Tensor) -> Tensor:
    """losses.py: L1 (:147-158), LogL1 (:161-174), torch.nn.MSELoss — on the VALID pixels (dn_model.py:731-736)."""
    if kind == "L1":
        return torch.abs(pred - gt).mean()
    if kind == "LogL1":
        return torch.log(1 + torch.abs(pred - gt)).mean()
    if kind == "MSE":
        return ((pred - gt) ** 2).mean()
    if kind == "HuberL1":
        # losses.py:217-238 as called with the valid pixels only (1-D tensors): the threshold is a fifth of the LARGEST
        # error of the frame, below it the error is quadratic
        m = gt != 0
        l1 = torch.abs(pred[m] - gt[m])
        d = 0.2 * torch.max(l1)
        return torch.where(l1 < d, ((pred - gt) ** 2 + d ** 2) / (2 * d), l1).mean()
    raise ValueError(kind)


def sparse_opacity_term(opacity_logits: Tensor, vis_indices: Tensor, step: int, sparse_loss_steps: int = 10,
                        reset_alpha_every: int = 30, refine_every: int = 100, sparse_lambda: float = 0.1) -> Tensor:
    """dn_model.py:821-837: the binary entropy of the VISIBLE Gaussians' opacities, every ``sparse_loss_steps``-th
    step, except on an opacity reset's step and during the 100 steps behind it."""
    zero = torch.zeros((), dtype=opacity_logits.dtype)
    if step % sparse_loss_steps != 0:
        return zero
    r = step % (reset_alpha_every * refine_every)
    if r == 0 or 1 <= r <= 100:
        return zero
    o = torch.sigmoid(opacity_logits[vis_indices])
    return sparse_lambda * (-o * torch.log(o + 1e-10) - (1 - o) * torch.log(1 - o + 1e-10)).mean()


def dn_terms(outputs: Dict[str, Tensor], batch: Dict[str, Tensor], log_scales: Tensor,
             normals_world: Optional[Tensor], add_mask: Optional[Tensor], touch_normals: Optional[Tensor],
             cfg: LossConfig, intrinsics=None) -> Tensor:
    """What dn_model.py:673-925 adds to splatfacto's main loss.  ``intrinsics`` = (fx, fy, cx, cy) of the camera of the
    last get_outputs, read by ``normal_supervision == "depth"`` (:775-786)."""
    gt_img = batch["image"].clamp(min=10 / 255.0)                      # :692
    depth_out = outputs["depth"]
    sensor = batch.get("sensor_depth")
    mono = batch.get("mono_depth")
    normal_gt = batch.get("normal")
    pred_normal = outputs["normal"]
    if "mask" in batch:                                                # :702-714
        mask = batch["mask"]
        depth_out = depth_out * mask
        if sensor is not None:
            sensor = sensor * mask
        if mono is not None:
            mono = mono * mask
        if normal_gt is not None:
            normal_gt = normal_gt * mask
        pred_normal = pred_normal * mask
    depth_loss = 0
    edge = cfg.depth_loss_type == "EdgeAwareLogL1"
    if cfg.use_depth_loss and sensor is not None and cfg.sensor_depth_lambda > 0.0:   # :720-736
        valid = sensor > cfg.depth_tolerance
        term = (edge_aware_log_l1(depth_out, sensor.float(), gt_img, valid) if edge else
                plain_depth_loss(cfg.depth_loss_type, depth_out[valid], sensor[valid].float()))
        depth_loss = depth_loss + cfg.sensor_depth_lambda * term
    if cfg.use_depth_loss and mono is not None and cfg.mono_depth_lambda > 0.0:       # :737-750
        valid = mono > (cfg.depth_tolerance if edge else 0.0)          # (the plain types test > 0, :739)
        term = (edge_aware_log_l1(depth_out, mono.float(), gt_img, valid) if edge else
                plain_depth_loss(cfg.depth_loss_type, depth_out[valid], mono[valid].float()))
        depth_loss = depth_loss + cfg.mono_depth_lambda * term
    if cfg.use_depth_smooth_loss:                                      # :752-760
        sm = tv(depth_out) if cfg.smooth_loss_type == "TV" else edge_aware_tv(depth_out, gt_img)
        depth_loss = depth_loss + cfg.smooth_loss_lambda * sm
    normal_loss = 0
    if cfg.use_normal_loss:                                            # :770-815
        if cfg.normal_supervision == "depth":                          # :774-795: pseudo normals of the (masked) depth
            fx, fy, cx, cy = intrinsics
            gt_normal = normal_from_depth(depth_out.detach(), fx, fy, cx, cy)
            gt_normal = gt_normal * torch.tensor([1.0, -1.0, -1.0], dtype=gt_normal.dtype)
            gt_normal = (1 + gt_normal) / 2
        else:
            gt_normal = normal_gt
        normal_loss = normal_loss + torch.abs(gt_normal - pred_normal).mean()
        if cfg.use_normal_cosine_loss:                                 # :807-813
            normal_loss = normal_loss + mean_angular_error_as_called(pred_normal, gt_normal)
        if cfg.use_normal_tv_loss:
            normal_loss = normal_loss + tv(pred_normal)
    if cfg.two_d_gaussians:                                            # :817-819
        normal_loss = normal_loss + torch.min(torch.exp(log_scales), dim=1, keepdim=True)[0].mean()
    total = depth_loss + cfg.normal_lambda * normal_loss               # :880-886
    if add_mask is not None:                                           # :893-902
        # self.normals is the gauss_params["normals"] entry written by the last get_outputs: a detached leaf
        # (ParameterDict re-wraps it), so this term moves the loss value and no trained parameter
        nt = normals_world.detach()[add_mask]
        assert nt.shape == touch_normals.shape
        total = total + torch.mean((nt - touch_normals) ** 2) * cfg.touch_normal_loss_lambda
    return total


def get_loss_dict(outputs: Dict[str, Tensor], batch: Dict[str, Tensor], log_scales: Tensor,
                  normals_world: Optional[Tensor] = None, add_mask: Optional[Tensor] = None,
                  touch_normals: Optional[Tensor] = None, cfg: Optional[LossConfig] = None,
                  intrinsics=None) -> Dict[str, Tensor]:
    """{"main_loss", "scale_reg"} as the reference returns them (nerfstudio's Trainer sums the values)."""
    cfg = cfg or LossConfig()
    main = photometric_loss(outputs["rgb"], batch["image"], batch.get("mask"), cfg.ssim_lambda)
    main = main + dn_terms(outputs, batch, log_scales, normals_world, add_mask, touch_normals, cfg, intrinsics)
    return {"main_loss": main, "scale_reg": torch.zeros((), dtype=main.dtype)}
