"""The legacy (gsplat 0.1.x, shipped in 1.0.0 as ``gsplat.cuda_legacy``) part of the operator
surface FusionSense imports (/root/reference/dn_splatter/dn_model.py:33-35):

* ``rasterize_gaussians``  — call site dn_model.py:644-653 (per-pixel normals, 3 channels)
* ``quat_to_rotmat``       — dn_model.py:623,1191,1489,1699,1770,2146
* ``num_sh_bases``         — dn_model.py:205,286
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from . import frame_cache, ops
from ._lib import require_gpu_tensor


def num_sh_bases(degree: int) -> int:
    """(degree+1)^2 for degree 0..4 (gsplat.cuda_legacy._wrapper.num_sh_bases)."""
    if degree == 0:
        return 1
    if degree == 1:
        return 4
    if degree == 2:
        return 9
    if degree == 3:
        return 16
    return 25


def quat_to_rotmat(quat: Tensor) -> Tensor:
    """wxyz quaternion(s) [...,4] -> rotation matrices [...,3,3]; the input is normalised first
    (gsplat.cuda_legacy._torch_impl.quat_to_rotmat).  Plain torch: it is a tiny helper used in
    initialisation / densification code on either device, not on the per-pixel path."""
    assert quat.shape[-1] == 4, quat.shape
    w, x, y, z = torch.unbind(F.normalize(quat, dim=-1), dim=-1)
    mat = torch.stack(
        [
            1 - 2 * (y**2 + z**2), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x**2 + z**2), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x**2 + y**2),
        ],
        dim=-1,
    )
    return mat.reshape(quat.shape[:-1] + (3, 3))


def rasterize_gaussians(
    xys: Tensor,  # [N,2]
    depths: Tensor,  # [N]
    radii: Tensor,  # [N] int32
    conics: Tensor,  # [N,3]
    num_tiles_hit: Tensor,  # [N] int32
    colors: Tensor,  # [N,ch]
    opacity: Tensor,  # [N,1]
    img_height: int,
    img_width: int,
    block_width: int,
    background: Optional[Tensor] = None,
    return_alpha: bool = False,
):
    """N-channel splat of per-Gaussian ``colors`` over the legacy binning.  Returns
    ``out_img [H,W,ch]`` (and ``out_alpha [H,W]`` if ``return_alpha``).

    Deviation (DESIGN.md §"legacy binning"): the reference sizes its intersection buffers from
    ``num_tiles_hit`` (counted with the new floor/ceil rule) but fills them with the legacy
    trunc/+1 rule, which overflows when ``mean + radius`` is an exact multiple of the tile
    size; here the legacy rule is used for both count and fill, so ``num_tiles_hit`` is only
    shape-checked.  Ties in (tile, depth) are kept in ascending Gaussian order (stable)."""
    from . import rendering as _rendering
    if _rendering.MEASURE_HOST_TIME:
        import time as _time
        _t0 = _time.perf_counter()
    if not (1 < block_width <= 16):
        raise AssertionError("block_width must be between 2 and 16")
    if colors.dtype == torch.uint8:
        colors = colors.float() / 255
    for name, t in (("xys", xys), ("depths", depths), ("conics", conics), ("colors", colors),
                    ("opacity", opacity)):
        require_gpu_tensor(t, name)
    require_gpu_tensor(radii, "radii", torch.int32)
    N = xys.shape[0]
    if xys.dim() != 2 or xys.shape[1] != 2:
        raise ValueError("xys must have dimensions (N, 2)")
    if colors.dim() != 2 or colors.shape[0] != N:
        raise ValueError("colors must have dimensions (N, D)")
    if num_tiles_hit.shape[0] != N or depths.shape[0] != N or conics.shape != (N, 3):
        raise ValueError("inconsistent N across inputs")
    ch = colors.shape[-1]
    dev = xys.device
    if background is not None:
        assert background.shape[0] == ch, f"incorrect shape of background color tensor, expected shape {ch}"
        background = background.to(device=dev, dtype=torch.float32)
    else:
        background = torch.ones(ch, dtype=torch.float32, device=dev)

    tw = (img_width + block_width - 1) // block_width
    th = (img_height + block_width - 1) // block_width
    cached = frame_cache.lookup(xys, depths, radii, img_width, img_height, block_width)
    live_payload = None
    on_dev = False
    if cached is not None:
        offsets, flatten_ids, isect_ids, live_payload, on_dev = cached  # same frame: reuse the first pass's sorted lists
    else:
        with torch.no_grad():
            _, isect_ids, flatten_ids, offsets, _ = ops.bin_and_sort(
                xys.detach()[None].contiguous(), radii[None].contiguous(), depths.detach()[None].contiguous(),
                block_width, tw, th, legacy=True)

    pad = 0
    cols = colors
    bg = background
    if ch == 2:
        pad = 1
    elif ch > 4:
        raise NotImplementedError("rasterize_gaussians supports 1..4 channels")
    if pad:
        cols = torch.cat([cols, torch.zeros(N, pad, device=dev)], dim=-1)
        bg = torch.cat([bg, torch.zeros(pad, device=dev)])
    out, alpha, _ = ops._Rasterize.apply(
        xys[None], conics[None], cols[None], opacity.reshape(1, N), bg[None], offsets, flatten_ids,
        int(img_width), int(img_height), int(block_width), False, isect_ids, False, live_payload, on_dev)
    out = out[0, ..., :ch]
    if _rendering.MEASURE_HOST_TIME:
        _rendering.HOST_TIME["rasterize_gaussians_s"] += _time.perf_counter() - _t0
        _rendering.HOST_TIME["rasterize_gaussians_calls"] += 1
    if return_alpha:
        return out, alpha[0, ..., 0]
    return out
