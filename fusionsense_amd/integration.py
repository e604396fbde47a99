"""Optional second step of the integration: ``DNSplatterModel.get_outputs`` itself on the fused node.

``gsplat_shim.install()`` keeps FusionSense's model file byte-for-byte and swaps the rasterizer underneath its two
calls; that route is bounded by the reference's own call structure (two compositing passes, ~60 small torch launches
of glue per iteration).  This module offers the same method — /root/reference/dn_splatter/dn_model.py:469-671, same
inputs, same returned dictionary, same side effects on the model object — computed by the fused node of
``fusionsense_amd/fused.py`` (one compositing walk for RGB+ED and the normal plane, no torch glue), still on torch's
autograd tape and under nerfstudio's own optimizers and callbacks:

    import dn_splatter.dn_model as m
    from fusionsense_amd import integration
    integration.patch(m.DNSplatterModel)          # get_outputs -> get_outputs_fused (the original stays as
                                                  # _get_outputs_reference and is used for what the node does not cover)

Side effects reproduced (they are what after_train / refinement_after / get_loss_dict read, dn_model.py:592-600,
607, 634, 660-663): ``self.xys`` (an object whose ``.absgrad`` [1,N,2] is filled by the backward), ``self.radii``,
``self.depths``, ``self.conics``, ``self.num_tiles_hit``, ``self.vis_indices``, ``self.last_size``,
``self.gauss_params["normals"]``, ``self.camera``, ``self.camera_idx``.
"""
from __future__ import annotations

from typing import Dict

import torch

from .scenes import Camera


def _binary_threshold(cfg, step: int):
    """dn_model.py:492-503: is the binary-opacity write performed at this step?"""
    if getattr(cfg, "use_binary_opacities", False) and step > cfg.warmup_length:
        skip_steps = cfg.reset_alpha_every * cfg.refine_every
        if not step % skip_steps == 0 and step % skip_steps not in range(1, 200 + 1):
            return float(cfg.binary_opacities_threshold)
    return None


def get_outputs_fused(self, camera) -> Dict[str, torch.Tensor]:
    """Drop-in for ``DNSplatterModel.get_outputs(camera)``."""
    from .fused import render_fusionsense_fused
    if not hasattr(camera, "camera_to_worlds"):
        print("Called get_outputs with not a camera")
        return {}
    cfg = self.config
    if cfg.sh_degree <= 0 or not cfg.predict_normals or cfg.rasterize_mode != "classic":
        return self._get_outputs_reference(camera)  # (sigmoid colours / no normals / antialiased: the op-by-op route)
    if self.training:
        assert camera.shape[0] == 1, "Only one camera at a time"
        c2w = self.camera_optimizer.apply_to_camera(camera)
    else:
        c2w = camera.camera_to_worlds
    if torch.is_tensor(c2w) and c2w.requires_grad:
        return self._get_outputs_reference(camera)  # (camera optimisation is off in FusionSense, dn_model.py:128-130)
    scale = self._get_downscale_factor()
    camera.rescale_output_resolution(1 / scale)
    K = camera.get_intrinsics_matrices()[0].detach().cpu()
    W, H = int(camera.width.item()), int(camera.height.item())
    camera.rescale_output_resolution(scale)
    self.last_size = (H, W)
    cam = Camera(c2w.reshape(-1, 3, 4)[0].detach().cpu().float(), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]),
                 float(K[1, 2]), W, H)
    dev = self.gauss_params["means"].device
    sh_degree_to_use = min(self.step // cfg.sh_degree_interval, cfg.sh_degree)
    background = self._get_background_color().to(device=dev, dtype=torch.float32)
    crop_box = self.crop_box if (self.crop_box is not None and not self.training) else None
    out = render_fusionsense_fused(self.gauss_params, cam, sh_degree=sh_degree_to_use, background=background, device=dev,
                                   add_mask=self.add_mask, crop_box=crop_box, training=bool(self.training),
                                   binary_threshold=_binary_threshold(cfg, self.step))
    if "info" not in out:  # the crop left nothing: get_empty_outputs
        return out
    info = out["info"]
    self.xys = info                      # .absgrad [1,N,2] after backward, as after_train reads it
    self.radii = info.radii[0]
    self.depths, self.conics, self.num_tiles_hit = info.depths, info.conics, info.tiles_per_gauss
    self.vis_indices = torch.where(self.radii > 0)[0]
    if crop_box is None:
        self.gauss_params["normals"] = out["normals_world"]
    md = getattr(camera, "metadata", None)
    if md is not None and "cam_idx" in md:
        self.camera_idx = md["cam_idx"]
    self.camera = camera
    return {"rgb": out["rgb"], "depth": out["depth"], "normal": out["normal"], "accumulation": out["accumulation"],
            "background": background}


def patch(model_cls) -> None:
    """Route ``model_cls.get_outputs`` through the fused node (idempotent)."""
    if getattr(model_cls, "_fsgs_patched", False):
        return
    model_cls._get_outputs_reference = model_cls.get_outputs
    model_cls.get_outputs = get_outputs_fused
    model_cls._fsgs_patched = True
