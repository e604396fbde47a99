"""Capacity estimates for the no-wait binning: how many live (Gaussian, tile) pairs to size a frame's
lists for before the frame's own count is known.

The estimate of a frame shape is the largest count that shape has shown over its last two windows of frames (so it
follows a model that shrinks, instead of growing for ever) plus a margin; shapes are keyed on a BUCKET of the Gaussian
count (quarter octaves), so the first frame after a densification — which changes N every ``refine_every`` steps —
finds its neighbour's estimate instead of falling back to the host wait, and the table is a small LRU instead of one
entry per N ever seen (rendering._LIVE_CAPS, trainer._live_caps).  A frame that outgrows its estimate is
redone with exact sizes by the caller (ops.LiveListOverflow) and raises the estimate at once."""
from __future__ import annotations

import math
from collections import OrderedDict


class LRU(OrderedDict):
    """A dictionary that forgets its least recently used entries beyond ``max_items``."""

    def __init__(self, max_items: int = 64):
        super().__init__()
        self.max_items = int(max_items)

    def get(self, key, default=None):
        if key in self:
            self.move_to_end(key)
            return OrderedDict.__getitem__(self, key)
        return default

    def __setitem__(self, key, value):
        OrderedDict.__setitem__(self, key, value)
        self.move_to_end(key)
        while len(self) > self.max_items:
            self.popitem(last=False)


def n_bucket(n: int) -> int:
    """Quarter-octave bucket of a Gaussian count (a 19 % step: inside the estimate's 25 % margin)."""
    return 0 if n <= 0 else int(math.ceil(4.0 * math.log2(n)))


class LiveCapacity:
    def __init__(self, max_keys: int = 64, window: int = 256, margin: float = 1.25, slack: int = 4096):
        self.table = LRU(max_keys)
        self.window, self.margin, self.slack = int(window), float(margin), int(slack)

    @staticmethod
    def key(dev, cams: int, n: int, width: int, height: int, extra=None):
        return (str(dev), int(cams), n_bucket(n), int(width), int(height), extra)

    def get(self, key) -> int:
        """Pairs to size the lists for, or 0 = unknown (the caller then waits for the frame's count)."""
        e = self.table.get(key)
        if e is None:
            return 0
        return int(max(e[0], e[1]) * self.margin) + self.slack

    def update(self, key, n_live: int) -> None:
        e = self.table.get(key)
        if e is None:
            e = self.table[key] = [0, 0, 0]  # (largest count of the running window, of the previous one, frames)
        e[0] = max(e[0], int(n_live))
        e[2] += 1
        if e[2] >= self.window:
            e[1], e[0], e[2] = e[0], int(n_live), 0

    def raise_to(self, key, needed: int) -> None:
        """After an overflow: the frame needed this many."""
        self.update(key, needed)

    def clear(self) -> None:
        self.table.clear()

    def __len__(self):
        return len(self.table)
