"""CPU oracle for the caller side: ``DNSplatterModel.get_outputs``
(/root/reference/dn_splatter/dn_model.py:469-671) written against oracle.gsplat_ref.
TEST INFRASTRUCTURE ONLY (see oracle/gsplat_ref.py header; parity unpinned)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from . import gsplat_ref as R


def render_fusionsense(gauss_params: Dict[str, Tensor], camera, sh_degree: int = 3,
                       background: Optional[Tensor] = None, predict_normals: bool = True,
                       rasterize_mode: str = "classic") -> Dict[str, Tensor]:
    means = gauss_params["means"]
    dt = means.dtype
    scales = gauss_params["scales"]
    quats = gauss_params["quats"]
    opacities = gauss_params["opacities"]
    colors = torch.cat((gauss_params["features_dc"][:, None, :], gauss_params["features_rest"]), dim=1)
    c2w = camera.c2w.to(dt)
    viewmat = R.get_viewmat(c2w[None])
    K = camera.K().to(dt)[None]
    W, H = camera.width, camera.height
    if background is None:
        background = torch.ones(3, dtype=dt)
    render, alpha, info = R.rasterization(
        means=means, quats=quats / quats.norm(dim=-1, keepdim=True), scales=torch.exp(scales),
        opacities=torch.sigmoid(opacities).squeeze(-1), colors=colors, viewmats=viewmat, Ks=K,
        width=W, height=H, tile_size=16, packed=False, near_plane=0.01, far_plane=1e10,
        render_mode="RGB+ED", sh_degree=sh_degree, sparse_grad=False, absgrad=True,
        rasterize_mode=rasterize_mode)
    xys = info["means2d"]
    radii = info["radii"][0]
    rgb = torch.clamp(render[:, ..., :3] + (1 - alpha) * background, 0.0, 1.0)
    depth_im = render[:, ..., 3:4]
    depth_im = torch.where(alpha > 0, depth_im, depth_im.detach().max()).squeeze(0)
    normals_im = torch.zeros_like(rgb.squeeze(0))
    normals_world = None
    if predict_normals:
        normals_world, normals = R.gaussian_normals(quats, scales, means, c2w)
        normals_im = R.rasterize_gaussians(
            xys[0].detach(), info["depths"][0], radii, info["conics"][0], info["tiles_per_gauss"][0],
            normals, torch.sigmoid(opacities), H, W, 16)
        normals_im = normals_im / normals_im.norm(dim=-1, keepdim=True)
        normals_im = (normals_im + 1) / 2
    return {"rgb": rgb.squeeze(0), "depth": depth_im, "normal": normals_im,
            "accumulation": alpha.squeeze(0), "background": background, "info": info, "xys": xys,
            "radii": radii, "normals_world": normals_world}
