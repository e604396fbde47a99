"""One-frame memo that lets ``rasterize_gaussians`` skip its own binning + sort.

FusionSense renders every frame twice (dn_splatter/dn_model.py:570-591, then :644-653): the
legacy call receives the SAME projected centres, depths and radii that ``rasterization`` just
binned and depth-sorted.  When the two bbox rules agree on every Gaussian of the frame
(fsgs_isect_count reports this) the sorted intersection lists are identical bit for bit, so
the second bin + 44-bit radix sort (156 B/intersection) is pure repetition.  The memo is keyed on
storage identity and version counters of the tensors, so any in-place edit or a different
tensor falls back to the full legacy binning."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

_frame = None
enabled = True
hits = 0
misses = 0


def _key(t: torch.Tensor) -> Tuple:
    return (t.data_ptr(), t._version, tuple(t.shape[-1:]), t.device)


def remember(means2d, depths, radii, width, height, tile_size, isect_offsets, flatten_ids,
             lists_valid_for_legacy: bool) -> None:
    global _frame
    if not lists_valid_for_legacy:
        _frame = None
        return
    _frame = dict(xys=_key(means2d), depths=_key(depths), radii=_key(radii), n=radii.shape[-1],
                  dims=(int(width), int(height), int(tile_size)), offsets=isect_offsets, flat=flatten_ids)


def lookup(xys, depths, radii, width, height, tile_size) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
    """(isect_offsets [1,th,tw], flatten_ids [M]) of the remembered frame, or None."""
    global hits, misses
    f = _frame
    if (enabled and f is not None and f["dims"] == (int(width), int(height), int(tile_size))
            and xys.shape[0] == f["n"] and _key(xys) == f["xys"] and _key(depths) == f["depths"]
            and _key(radii) == f["radii"]):
        hits += 1
        return f["offsets"], f["flat"]
    misses += 1
    return None


def clear() -> None:
    global _frame
    _frame = None
