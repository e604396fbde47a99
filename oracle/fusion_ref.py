"""CPU oracle for the caller side: ``DNSplatterModel.get_outputs``
(/root/reference/dn_splatter/dn_model.py:469-671) written against oracle.gsplat_ref.
TEST INFRASTRUCTURE ONLY (see oracle/gsplat_ref.py header; parity unpinned)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from . import gsplat_ref as R


def binary_opacity_step(step: int, warmup_length: int = 500, reset_alpha_every: int = 30, refine_every: int = 100,
                        use_binary_opacities: bool = True) -> bool:
    """dn_model.py:492-499, transcribed: is the binary-opacity write performed at this step?"""
    if use_binary_opacities and step > warmup_length:
        skip_steps = reset_alpha_every * refine_every
        margin = 200
        if not step % skip_steps == 0 and step % skip_steps not in range(1, margin + 1):
            return True
    return False


def obb_within(R3: Tensor, T: Tensor, S: Tensor, pts: Tensor) -> Tensor:
    """nerfstudio 1.1.3 OrientedBox.within (recalled): homogeneous world->box transform, open interval."""
    Hm = torch.eye(4, dtype=pts.dtype)
    Hm[:3, :3] = R3.to(pts.dtype)
    Hm[:3, 3] = T.to(pts.dtype)
    w2b = torch.linalg.inv(Hm)
    ph = torch.cat((pts, torch.ones_like(pts[..., :1])), dim=-1)
    loc = (w2b @ ph.T).T[..., :3]
    Sd = S.to(pts.dtype)
    return torch.all(torch.cat([loc > -Sd / 2, loc < Sd / 2], dim=-1), dim=-1)


def render_fusionsense(gauss_params: Dict[str, Tensor], camera, sh_degree: int = 3,
                       background: Optional[Tensor] = None, predict_normals: bool = True,
                       rasterize_mode: str = "classic", add_mask: Optional[Tensor] = None,
                       crop_box=None, training: bool = True, binary_threshold: Optional[float] = None
                       ) -> Dict[str, Tensor]:
    """dn_model.py:469-671.  ``binary_threshold`` (when the schedule says so, :492-503) rewrites the opacity
    parameter's data; ``crop_box`` = (R, T, S) of an OrientedBox, used only when not training (:505-532);
    ``add_mask`` marks touch anchors whose means / opacities / scales are detached (:535-541)."""
    dt = gauss_params["means"].dtype
    if binary_threshold is not None:
        o = gauss_params["opacities"]
        o.data = torch.where(o.data >= binary_threshold, torch.ones_like(o.data), torch.zeros_like(o.data))
    if background is None:
        background = torch.ones(3, dtype=dt)
    if crop_box is not None and not training:
        crop_ids = obb_within(crop_box[0], crop_box[1], crop_box[2], gauss_params["means"].detach())
        if crop_ids.sum() == 0:
            H, W = camera.height, camera.width
            return {"rgb": background.repeat(H, W, 1), "depth": background.new_ones(H, W, 1) * 10,
                    "accumulation": background.new_zeros(H, W, 1), "background": background}
        gauss_params = {k: v[crop_ids] for k, v in gauss_params.items()}
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
    c2w = camera.c2w.to(dt)
    viewmat = R.get_viewmat(c2w[None])
    K = camera.K().to(dt)[None]
    W, H = camera.width, camera.height
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
