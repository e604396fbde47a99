"""Host-side mirror of ``DNSplatterModel.get_outputs``
(/root/reference/dn_splatter/dn_model.py:469-671): the exact argument preparation FusionSense
performs around the two rasterizer calls, so that tests and the benchmark exercise the boundary
the way the reference does.  All heavy lifting goes through :mod:`fusionsense_amd.rendering`
and :mod:`fusionsense_amd.legacy` (i.e. libfsgs.so).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from . import ops
from .legacy import rasterize_gaussians
from .rendering import rasterization
from .scenes import Camera

BLOCK_WIDTH = 16  # dn_model.py:545-547


def get_viewmat(optimized_camera_to_world: Tensor) -> Tensor:
    """nerfstudio.models.splatfacto.get_viewmat (used at dn_model.py:550): OpenGL c2w [B,3,4]
    -> OpenCV world-to-camera [B,4,4].  Tiny host-side torch math."""
    R = optimized_camera_to_world[:, :3, :3]
    T = optimized_camera_to_world[:, :3, 3:4]
    R = R * torch.tensor([[[1.0, -1.0, -1.0]]], device=R.device, dtype=R.dtype)
    R_inv = R.transpose(1, 2)
    T_inv = -torch.bmm(R_inv, T)
    viewmat = torch.zeros(R.shape[0], 4, 4, device=R.device, dtype=R.dtype)
    viewmat[:, 3, 3] = 1.0
    viewmat[:, :3, :3] = R_inv
    viewmat[:, :3, 3:4] = T_inv
    return viewmat


def render_fusionsense(
    gauss_params: Dict[str, Tensor],
    camera: Camera,
    sh_degree: int = 3,
    background: Optional[Tensor] = None,
    predict_normals: bool = True,
    rasterize_mode: str = "classic",
    device: Optional[torch.device] = None,
    fused_normals: bool = True,
    add_mask: Optional[Tensor] = None,
    crop_box=None,
    training: bool = True,
    binary_threshold: Optional[float] = None,
) -> Dict[str, Tensor]:
    """One ``get_outputs`` call.  ``gauss_params`` uses the reference's stored parametrisation
    (dn_model.py:294-304): means, scales (log), quats, features_dc [N,3], features_rest [N,K-1,3],
    opacities (logit) [N,1].  ``sh_degree`` is ``sh_degree_to_use`` (dn_model.py:562-568).
    ``binary_threshold``: perform the binary-opacity write of :492-503 (the caller decides with
    ``splatfacto.binary_opacity_active``); ``crop_box`` (:class:`fusionsense_amd.crop.OrientedBox`) is used
    only when not ``training`` (:505-532); ``add_mask`` [N] bool marks the touch anchors whose means /
    opacities / scales receive no gradient (:535-541)."""
    dev = device or gauss_params["means"].device
    if binary_threshold is not None:
        from .splatfacto import binary_opacity_write_
        binary_opacity_write_(gauss_params["opacities"], binary_threshold)
    if crop_box is not None and not training:
        from .crop import crop_params, get_empty_outputs
        crop_ids = crop_box.within(gauss_params["means"].detach())
        if crop_ids.sum() == 0:
            bg = background if background is not None else torch.ones(3, device=dev)
            return get_empty_outputs(camera.width, camera.height, bg)
        gauss_params = crop_params(gauss_params, crop_ids)
    means = gauss_params["means"]
    scales = gauss_params["scales"]
    quats = gauss_params["quats"]
    opacities = gauss_params["opacities"]
    if add_mask is not None:
        opacities = opacities.clone()
        opacities[add_mask] = opacities[add_mask].detach()
        means = means.clone()
        means[add_mask] = means[add_mask].detach()
        scales = scales.clone()
        scales[add_mask] = scales[add_mask].detach()
    colors = torch.cat((gauss_params["features_dc"][:, None, :], gauss_params["features_rest"]), dim=1)
    c2w = camera.c2w.to(dev)
    viewmat = get_viewmat(c2w[None])
    K = camera.K().to(dev)[None]
    W, H = camera.width, camera.height
    if background is None:
        background = torch.ones(3, device=dev)  # background_color="white" (dn_model.py:141)

    render, alpha, info = rasterization(
        means=means,
        quats=quats / quats.norm(dim=-1, keepdim=True),
        scales=torch.exp(scales),
        opacities=torch.sigmoid(opacities).squeeze(-1),
        colors=colors,
        viewmats=viewmat,
        Ks=K,
        width=W,
        height=H,
        tile_size=BLOCK_WIDTH,
        packed=False,
        near_plane=0.01,
        far_plane=1e10,
        render_mode="RGB+ED",
        sh_degree=sh_degree,
        sparse_grad=False,
        absgrad=True,
        rasterize_mode=rasterize_mode,
    )
    if info["means2d"].requires_grad:
        info["means2d"].retain_grad()
    xys = info["means2d"]
    radii = info["radii"][0]
    rgb = render[:, ..., :3] + (1 - alpha) * background
    rgb = torch.clamp(rgb, 0.0, 1.0)
    depth_im = render[:, ..., 3:4]
    depth_im = torch.where(alpha > 0, depth_im, depth_im.detach().max()).squeeze(0)

    normals_im = torch.zeros_like(rgb.squeeze(0))
    normals_world = None
    if predict_normals:
        if fused_normals:
            normals_world, normals = ops.gaussian_normals(quats, scales, means, c2w)
        else:  # the reference's op-by-op torch formulation, kept for A/B timing
            q = quats / quats.norm(dim=-1, keepdim=True)
            from .legacy import quat_to_rotmat
            oh = torch.nn.functional.one_hot(torch.argmin(scales, dim=-1), num_classes=3).float()
            n = torch.bmm(quat_to_rotmat(q), oh[:, :, None]).squeeze(-1)
            n = torch.nn.functional.normalize(n, dim=1)
            vd = -means.detach() + c2w.detach()[:3, 3]
            vd = vd / vd.norm(dim=-1, keepdim=True)
            n = torch.where(((n * vd).sum(-1) < 0)[:, None], -n, n)
            normals_world = n
            normals = n @ c2w[:3, :3]
        normals_im = rasterize_gaussians(
            xys[0, ...].detach(),
            info["depths"][0, ...],
            radii,
            info["conics"][0, ...],
            info["tiles_per_gauss"][0, ...],
            normals,
            torch.sigmoid(opacities),
            H, W, BLOCK_WIDTH,
        )
        normals_im = normals_im / normals_im.norm(dim=-1, keepdim=True)
        normals_im = (normals_im + 1) / 2
    return {
        "rgb": rgb.squeeze(0),
        "depth": depth_im,
        "normal": normals_im,
        "accumulation": alpha.squeeze(0),
        "background": background,
        "info": info,
        "xys": xys,
        "radii": radii,
        "normals_world": normals_world,
    }
