"""The eval-time crop of ``get_outputs`` (dn_model.py:505-532) and what it returns when nothing is left.

``OrientedBox`` restates nerfstudio 1.1.3 ``nerfstudio.data.scene_box.OrientedBox.within`` (un-vendored;
recalled): a point is inside iff its coordinates in the box frame, ``inv([R|T]) p``, lie
strictly between ``-S/2`` and ``S/2``.  ``get_empty_outputs`` restates ``SplatfactoModel.get_empty_outputs``:
the background colour everywhere, depth 10, zero accumulation.  FusionSense reaches both through
``get_outputs_for_camera(camera, obb_box)`` (scripts/render_video.py:239-246, export_mesh.py:355)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import torch
from torch import Tensor


@dataclass
class OrientedBox:
    R: Tensor  # [3,3] box-to-world rotation
    T: Tensor  # [3] centre
    S: Tensor  # [3] edge lengths

    def within(self, pts: Tensor) -> Tensor:
        """[N,3] -> bool [N]."""
        Rm, T, S = self.R.to(pts), self.T.to(pts), self.S.to(pts)
        local = (pts - T) @ Rm  # rows: R^T (p - T)  ==  inv([R|T]) p for a rotation R
        return ((local > -S / 2) & (local < S / 2)).all(dim=-1)


def get_empty_outputs(width: int, height: int, background: Tensor) -> Dict[str, Tensor]:
    rgb = background.repeat(height, width, 1)
    depth = background.new_ones(height, width, 1) * 10
    accumulation = background.new_zeros(height, width, 1)
    return {"rgb": rgb, "depth": depth, "accumulation": accumulation, "background": background}


def crop_params(gauss_params: Dict[str, Tensor], crop_ids: Tensor) -> Dict[str, Tensor]:
    """The six ``*_crop`` gathers of dn_model.py:519-525."""
    return {k: v[crop_ids] for k, v in gauss_params.items()}
