"""``gsplat.rendering.rasterization`` re-implemented on libfsgs.so.

Drop-in for the call at /root/reference/dn_splatter/dn_model.py:570-591: same name, argument
meaning, return triple and ``meta`` keys as gsplat 1.0.0.  FusionSense passes
``packed=False, render_mode="RGB+ED", absgrad=True, sparse_grad=False, tile_size=16``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import frame_cache, ops
from ._lib import require_gpu_tensor

_RENDER_MODES = ("RGB", "D", "ED", "RGB+D", "RGB+ED")

# No host wait inside rasterization() either (round 3; the tape-free step has had it since round 2): the live lists
# are sized from the largest live-pair count this (device, C, N, image) shape has shown so far (+ 25 %), the kernels
# take the lists' true end from the device, and the host looks at the total only after the compositing forward is
# enqueued; a frame that exceeds the estimate is binned and composited once more with exact sizes (forward only: nothing
# to undo).  FSGS_NO_WAIT=0 restores the wait in front of the bucket fill.
import os as _os
NO_WAIT = _os.environ.get("FSGS_NO_WAIT", "1") != "0"
from .capacity import LiveCapacity as _LiveCapacity
_LIVE_CAPS = _LiveCapacity()  # (bucketed N, windowed maximum, LRU-bounded: capacity.py)
live_overflows = 0

# host seconds spent inside rasterization() / rasterize_gaussians() (library side of the drop-in route) and calls,
# for bench.py's dropin_host_ms_per_call — only while MEASURE_HOST_TIME is set (bench.py does, around that side
# measurement; FSGS_MEASURE_HOST_TIME=1 from the environment): the product call carries no clock reads otherwise
MEASURE_HOST_TIME = _os.environ.get("FSGS_MEASURE_HOST_TIME", "0") == "1"
HOST_TIME = {"rasterization_s": 0.0, "rasterization_calls": 0, "rasterize_gaussians_s": 0.0, "rasterize_gaussians_calls": 0}


class LazyMeta(dict):
    """gsplat's ``meta`` dictionary whose three list entries — ``isect_ids``, ``flatten_ids``, ``isect_offsets``:
    gsplat's FULL bounding-box tile lists, 12 B per intersection and six radix passes to sort — are built the first
    time someone asks for them.  FusionSense never does (dn_model.py:592-600 reads means2d, radii, depths, conics,
    tiles_per_gauss), and the compositing itself walks the frame's LIVE lists (only pairs that can reach a pixel),
    which give the same images, alphas and gradients."""
    LAZY = ("isect_ids", "flatten_ids", "isect_offsets", "legacy_rule_diff")

    def __init__(self, eager: dict, build):
        super().__init__(eager)
        self._build_lists = build

    def _materialise(self):
        if self._build_lists is not None:
            build, self._build_lists = self._build_lists, None
            self.update(build())

    def __missing__(self, key):
        if key in self.LAZY and self._build_lists is not None:
            self._materialise()
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or (key in self.LAZY and self._build_lists is not None)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def keys(self):
        self._materialise()
        return dict.keys(self)

    def items(self):
        self._materialise()
        return dict.items(self)

    def values(self):
        self._materialise()
        return dict.values(self)

    def __iter__(self):
        self._materialise()
        return dict.__iter__(self)

    def __len__(self):
        self._materialise()
        return dict.__len__(self)

    # C-level dict paths (dict.copy, dict(meta), {**meta}, dict.update(other, meta), pickling, copy.copy) do not go
    # through the overrides above: everything that hands the CONTENT on materialises the lists first and returns a
    # plain dict, so a copy can never lack a key that ``in`` reported.
    def copy(self):
        self._materialise()
        return dict(dict.items(self))

    __copy__ = copy

    def __deepcopy__(self, memo):
        import copy as _copy
        self._materialise()
        return {k: _copy.deepcopy(v, memo) for k, v in dict.items(self)}

    def __reduce__(self):
        self._materialise()
        return (dict, (dict(dict.items(self)),))

    def __or__(self, other):
        return self.copy() | dict(other)

    def __ror__(self, other):
        return dict(other) | self.copy()


def _validate(means, quats, scales, opacities, colors, viewmats, Ks, sh_degree, backgrounds, render_mode,
              tile_size, rasterize_mode):
    for name, t in (("means", means), ("quats", quats), ("scales", scales), ("opacities", opacities),
                    ("colors", colors), ("viewmats", viewmats), ("Ks", Ks)):
        require_gpu_tensor(t, name)
    N = means.shape[0]
    C = viewmats.shape[0]
    if means.shape != (N, 3):
        raise ValueError(f"means must be [N,3], got {tuple(means.shape)}")
    if quats.shape != (N, 4):
        raise ValueError(f"quats must be [N,4], got {tuple(quats.shape)}")
    if scales.shape != (N, 3):
        raise ValueError(f"scales must be [N,3], got {tuple(scales.shape)}")
    if opacities.shape != (N,):
        raise ValueError(f"opacities must be [N], got {tuple(opacities.shape)}")
    if viewmats.shape != (C, 4, 4):
        raise ValueError(f"viewmats must be [C,4,4], got {tuple(viewmats.shape)}")
    if Ks.shape != (C, 3, 3):
        raise ValueError(f"Ks must be [C,3,3], got {tuple(Ks.shape)}")
    if render_mode not in _RENDER_MODES:
        raise ValueError(f"render_mode must be one of {_RENDER_MODES}, got {render_mode!r}")
    if rasterize_mode not in ("classic", "antialiased"):
        raise ValueError(f"rasterize_mode must be 'classic' or 'antialiased', got {rasterize_mode!r}")
    if not (2 <= tile_size <= 16):
        raise ValueError("tile_size must be in [2,16]")
    if sh_degree is None:
        if not ((colors.dim() == 2 and colors.shape[0] == N) or
                (colors.dim() == 3 and colors.shape[:2] == (C, N))):
            raise ValueError(f"colors must be [N,D] or [C,N,D], got {tuple(colors.shape)}")
    else:
        if not (colors.dim() == 3 and colors.shape[0] == N and colors.shape[2] == 3):
            raise ValueError(f"colors must be SH coefficients [N,K,3], got {tuple(colors.shape)}")
        if (sh_degree + 1) ** 2 > colors.shape[1]:
            raise ValueError(f"sh_degree {sh_degree} needs {(sh_degree + 1) ** 2} bases, got {colors.shape[1]}")
        if not 0 <= sh_degree <= 4:
            raise ValueError("sh_degree must be in [0,4]")
    if backgrounds is not None:
        require_gpu_tensor(backgrounds, "backgrounds")
        if backgrounds.dim() != 2 or backgrounds.shape[0] != C:
            raise ValueError(f"backgrounds must be [C,D], got {tuple(backgrounds.shape)}")
    return N, C


def rasterization(
    means: Tensor,  # [N,3]
    quats: Tensor,  # [N,4] wxyz
    scales: Tensor,  # [N,3]
    opacities: Tensor,  # [N]
    colors: Tensor,  # [N,D] | [C,N,D] | SH [N,K,3]
    viewmats: Tensor,  # [C,4,4] world-to-camera (OpenCV)
    Ks: Tensor,  # [C,3,3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = True,
    tile_size: int = 16,
    backgrounds: Optional[Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    channel_chunk: int = 32,
) -> Tuple[Tensor, Tensor, Dict]:
    """Render C views of N Gaussians.  Returns (render [C,H,W,D'], alphas [C,H,W,1], meta).

    ``packed`` only changes gsplat's internal memory layout, never the result; this
    implementation always uses the dense ``[C,N,...]`` layout FusionSense asks for
    (``packed=False``) and reports ``camera_ids = gaussian_ids = None`` accordingly.
    ``sparse_grad`` is a packed-mode option and must be False."""
    if MEASURE_HOST_TIME:
        import time as _time
        _t0 = _time.perf_counter()
    if sparse_grad:
        raise ValueError("sparse_grad=True requires packed mode, which this backend does not expose")
    N, C = _validate(means, quats, scales, opacities, colors, viewmats, Ks, sh_degree, backgrounds,
                     render_mode, tile_size, rasterize_mode)
    width, height = int(width), int(height)
    dev = means.device
    antialiased = rasterize_mode == "antialiased"

    radii, means2d, depths, conics, comp = ops._Projection.apply(
        means, quats, scales, viewmats, Ks, width, height, float(eps2d), float(near_plane),
        float(far_plane), float(radius_clip), antialiased)
    opac = opacities[None, :].expand(C, N)
    if antialiased:
        opac = opac * comp

    tile_width = math.ceil(width / tile_size)
    tile_height = math.ceil(height / tile_size)
    lib = ops.load()
    # Fast path (tile 16): only the (Gaussian, tile) pairs that can reach a pixel are binned, straight into their
    # tile's bucket, and sorted per tile in LDS (ops.bin_live_*); the count's read-back overlaps the colour
    # evaluation below.  gsplat's full lists are built lazily, if the caller ever reads them from ``meta``.
    live_bins = (tile_size == 16 and ops.USE_LIVE_LISTS and ops.USE_BIN_LIVE
                 and C * tile_width * tile_height <= lib.fsgs_bin_live_max_tiles())
    count = None
    rule_diff = None
    cap_key = _LIVE_CAPS.key(dev, C, N, width, height)
    if live_bins:
        with torch.no_grad():
            opac_c = opac.detach().contiguous()
            count = ops.bin_live_count_async(means2d.detach(), radii, conics.detach(), opac_c, tile_width, tile_height,
                                             # (no Gaussians: nothing to gather from — an estimate left by frames of the
                                             # same shape with one Gaussian would announce pairs that have no records)
                                             capacity=_LIVE_CAPS.get(cap_key) if (NO_WAIT and N > 0) else 0)
    else:
        with torch.no_grad():
            tiles_per_gauss, isect_ids, flatten_ids, isect_offsets, rule_diff = ops.bin_and_sort(
                means2d, radii, depths, tile_size, tile_width, tile_height, legacy=False)
        # let the legacy normal pass (dn_model.py:644-653) reuse these lists when it is handed this
        # frame's own xys / depths / radii (see frame_cache for why that is the faithful choice even
        # for the rule_diff Gaussians whose bbox touches a tile edge exactly)
        frame_cache.remember(means2d, depths, radii, width, height, tile_size, isect_offsets, flatten_ids, isect_ids,
                             lists_valid_for_legacy=(C == 1))

    want_depth = render_mode in ("RGB+D", "RGB+ED")
    only_depth = render_mode in ("D", "ED")
    if only_depth:
        cols = depths[..., None]
        if backgrounds is not None:
            backgrounds = torch.zeros(C, 1, device=dev)
    elif sh_degree is None:
        cols = colors[None].expand(C, -1, -1) if colors.dim() == 2 else colors
        if want_depth:
            cols = torch.cat([cols, depths[..., None]], dim=-1)
    else:
        if viewmats.requires_grad:
            # differentiable camera pose (nerfstudio's camera optimiser; off in FusionSense, dn_model.py:128-130):
            # gsplat takes campos = inverse(viewmat)[:3, 3]; for [A t; 0 1] that is -A^-1 t, formed here with torch
            # ops so that autograd carries the SH view-direction gradient back into the view matrix
            if C != 1:
                raise NotImplementedError("SH colours with a differentiable camera pose: one camera per call")
            A, t = viewmats[:, :3, :3], viewmats[:, :3, 3:]
            campos = -ops.inverse3x3(A).matmul(t).squeeze(-1)
        else:
            campos = ops.campos_from_viewmats(viewmats)
        cols = ops._SHColors.apply(means, colors, campos, radii, depths if want_depth else None,
                                   int(sh_degree))
    if want_depth and backgrounds is not None:
        backgrounds = torch.cat([backgrounds, torch.zeros(C, 1, device=dev)], dim=-1)

    D = cols.shape[-1]
    expected_depth = render_mode in ("ED", "RGB+ED")
    fused_ed = expected_depth and D in (1, 3, 4) and tile_size == 16 and ops.USE_LIVE_LISTS
    live_payload = live_offsets = None
    T = C * tile_width * tile_height

    def bin_finish(cnt):
        """Second binning pass + in-tile sorts -> (tiles_per_gauss, payload, offsets [C,th,tw] view, ends_on_device)."""
        with torch.no_grad():
            tpg, _, pay, offs = ops.bin_live_finish(cnt, means2d.detach(), radii, depths.detach(), conics.detach(),
                                                    opac_c, tile_width, tile_height)
        on_dev = bool(cnt.get("capacity", 0)) and offs.dim() == 1  # (a dense frame's fallback chain waits and is exact)
        if on_dev:
            offs = offs[:T].view(C, tile_height, tile_width)  # (entry T of the storage behind it: the lists' end)
        frame_cache.remember(means2d, depths, radii, width, height, tile_size, offs, None, None,
                             lists_valid_for_legacy=(C == 1), live_payload=pay, ends_on_device=on_dev)
        return tpg, pay, offs, on_dev

    def composite(pay, offs, on_dev):
        if D in (1, 3, 4):
            return ops._Rasterize.apply(means2d, conics, cols, opac, backgrounds, offs, flatten_ids, width, height,
                                        tile_size, absgrad, isect_ids, fused_ed, pay, on_dev)
        # arbitrary channel counts: composite in chunks of <=4 channels over the same lists
        if absgrad:
            raise NotImplementedError("absgrad is defined per pixel over all channels; only D in {1,3,4}")
        outs = []
        alphas_ = last_ = None
        for s in range(0, D, 4):
            e = min(s + 4, D)
            w = e - s
            chunk = cols[..., s:e]
            bg = backgrounds[..., s:e] if backgrounds is not None else None
            if w == 2:
                chunk = torch.cat([chunk, torch.zeros_like(chunk[..., :1])], dim=-1)
                bg = torch.cat([bg, torch.zeros_like(bg[..., :1])], dim=-1) if bg is not None else None
            r, alphas_, last_ = ops._Rasterize.apply(
                means2d, conics, chunk.contiguous(), opac, bg, offs, flatten_ids, width, height,
                tile_size, False, isect_ids, False, pay, on_dev)
            outs.append(r[..., :w])
        return torch.cat(outs, dim=-1), alphas_, last_

    if count is not None:
        isect_ids = flatten_ids = None
        tiles_per_gauss, live_payload, live_offsets, on_dev = bin_finish(count)
        render, alphas, last_ids = composite(live_payload, live_offsets, on_dev)
        if on_dev:
            # the forward is enqueued: now look at the frame's live total (it has normally long arrived)
            global live_overflows
            try:
                n_live = ops.bin_live_check(count)
            except ops.LiveListOverflow as e:  # rare: once more with exact sizes (and the wait)
                live_overflows += 1
                n_live = e.needed
                with torch.no_grad():
                    count = ops.bin_live_count_async(means2d.detach(), radii, conics.detach(), opac_c, tile_width,
                                                     tile_height)
                tiles_per_gauss, live_payload, live_offsets, on_dev = bin_finish(count)
                render, alphas, last_ids = composite(live_payload, live_offsets, on_dev)
            live_payload = live_payload[:n_live]
        else:
            n_live = int(live_payload.numel())
        if NO_WAIT:
            _LIVE_CAPS.update(cap_key, n_live)
        isect_offsets = live_offsets
    else:
        render, alphas, last_ids = composite(None, isect_offsets, False)

    if expected_depth and not fused_ed:
        render = torch.cat([render[..., :-1], render[..., -1:] / alphas.clamp(min=1e-10)], dim=-1)

    eager = {
        "camera_ids": None,
        "gaussian_ids": None,
        "radii": radii,
        "means2d": means2d,
        "depths": depths,
        "conics": conics,
        "opacities": opac,
        "tile_width": tile_width,
        "tile_height": tile_height,
        "tiles_per_gauss": tiles_per_gauss,
        "width": width,
        "height": height,
        "tile_size": tile_size,
        "n_cameras": C,
        # extras (not in gsplat's meta).  last_ids index the list the compositing walked: the live list on the fast
        # path (live_payload / live_offsets), gsplat's full list otherwise
        "last_ids": last_ids,
        "live_payload": live_payload,
        "live_offsets": live_offsets,
    }
    if live_payload is None:
        eager.update(isect_ids=isect_ids, flatten_ids=flatten_ids, isect_offsets=isect_offsets,
                     legacy_rule_diff=rule_diff)  # Gaussians the legacy trunc/+1 bbox rule would bin differently
        meta = LazyMeta(eager, None)
    else:
        m2, rd, dp = means2d.detach(), radii, depths.detach()

        def build():
            with torch.no_grad():
                _, ids, flat, offs, diff = ops.bin_and_sort(m2, rd, dp, tile_size, tile_width, tile_height, legacy=False)
            return {"isect_ids": ids, "flatten_ids": flat, "isect_offsets": offs, "legacy_rule_diff": diff}

        meta = LazyMeta(eager, build)
    if MEASURE_HOST_TIME:
        HOST_TIME["rasterization_s"] += _time.perf_counter() - _t0
        HOST_TIME["rasterization_calls"] += 1
    return render, alphas, meta
