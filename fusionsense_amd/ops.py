"""Operator layer: thin Python wrappers + autograd Functions over the C-ABI (include/fsgs.h).

These are the operators gsplat 1.0.0 exposes under ``gsplat.cuda._wrapper`` and that
``gsplat.rendering.rasterization`` (reference call site dn_splatter/dn_model.py:570-591)
strings together.  torch provides device memory, the current stream and the autograd tape;
all arithmetic happens in libfsgs.so.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
import time
import math
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import check, load, ptr, stream_ptr


def _c(t: Optional[Tensor]) -> Optional[Tensor]:
    return None if t is None else t.contiguous()


class _KernelTimer:
    """HIP-event timing of the C-ABI launches on torch's current stream (the stream every kernel
    is enqueued on).  Disabled by default; bench.py enables it over its timed region."""

    def __init__(self):
        self.enabled = False
        self.records = []
        self.only = None
        self.pool = []
        self.every, self.seen = 1, 0

    def reset(self, enabled: bool = False, only=None, prealloc: int = 0, every: int = 1) -> None:
        """only: optional tuple of name prefixes; other launches are not timed (creating and recording two
        events per launch costs ~10 us of host time, i.e. a GPU bubble in front of the launch).  prealloc:
        number of spans whose events are created AND instantiated now, outside the region being timed."""
        self.enabled = enabled
        self.records = []
        self.only = tuple(only) if only else None
        self.every, self.seen = max(1, int(every)), 0  # (every: time each k-th of the selected launches only)
        self.pool = []
        for _ in range(2 * prealloc):
            e = torch.cuda.Event(enable_timing=True)
            e.record()  # (the HIP event is created lazily at its first record)
            self.pool.append(e)

    def _event(self):
        return self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)

    class _Span:
        def __init__(self, timer, name):
            self.timer, self.name = timer, name

        def __enter__(self):
            self.on = self.timer.enabled and (self.timer.only is None or self.name.startswith(self.timer.only))
            if self.on and self.timer.every > 1:
                self.timer.seen += 1
                self.on = (self.timer.seen - 1) % self.timer.every == 0
            if self.on:
                self.start = self.timer._event()
                self.end = self.timer._event()
                self.h0 = time.perf_counter()
                self.start.record()
            return self

        def __exit__(self, *exc):
            if self.on:
                self.end.record()
                self.timer.records.append((self.name, self.start, self.end, (time.perf_counter() - self.h0) * 1e3))
            return False

    def span(self, name: str):
        return _KernelTimer._Span(self, name)

    def summary(self):
        if not self.records:
            return {}
        torch.cuda.synchronize()
        acc = {}
        for name, s, e, host_ms in self.records:
            t = s.elapsed_time(e)
            a = acc.setdefault(name, [0.0, 0, t, t, 0.0])
            a[0] += t
            a[1] += 1
            a[2] = min(a[2], t)
            a[3] = max(a[3], t)
            a[4] = max(a[4], host_ms)
        return {k: {"avg_ms": round(v[0] / v[1], 4), "calls": v[1], "min_ms": round(v[2], 4), "max_ms": round(v[3], 4),
                    "host_max_ms": round(v[4], 4)} for k, v in sorted(acc.items())}


TIMER = _KernelTimer()


def _run(fn, args, what: str, tag: str = "") -> None:
    """Call one C-ABI entry point, time it if the timer is on, raise on a non-zero code."""
    # (the common case — timer off, or on for other launches than this one: no span object, no context manager)
    if not TIMER.enabled or (TIMER.only is not None and not what.startswith(TIMER.only, 5)):
        rc = fn(*args)
        if rc != 0:
            check(rc, what)
        return
    with TIMER.span(what[5:] + tag):
        rc = fn(*args)
    check(rc, what)


# dispatch order of the compositing backward on the drop-in route (an argument of the launch; the fused node applies
# its rule of the frame's size, fused._BwdDispatch): row-major unless FSGS_BWD_PERM=<stride> asks otherwise
DROPIN_BWD_STRIDE = int(os.environ.get("FSGS_BWD_PERM", "0")) if os.environ.get("FSGS_BWD_PERM", "auto") != "auto" else 0


def tile_bits(n_tiles: int) -> int:
    return int(math.floor(math.log2(n_tiles))) + 1 if n_tiles > 0 else 1


# --------------------------------------------------------------------------------------
# raw (non-differentiable) calls
# --------------------------------------------------------------------------------------
def project_fwd(means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
                radius_clip, calc_compensations=False):
    lib = load()
    N, Cn = means.shape[0], viewmats.shape[0]
    dev = means.device
    radii = torch.empty(Cn, N, dtype=torch.int32, device=dev)
    means2d = torch.empty(Cn, N, 2, dtype=torch.float32, device=dev)
    depths = torch.empty(Cn, N, dtype=torch.float32, device=dev)
    conics = torch.empty(Cn, N, 3, dtype=torch.float32, device=dev)
    comp = torch.empty(Cn, N, dtype=torch.float32, device=dev) if calc_compensations else None
    _run(lib.fsgs_project_fwd, (Cn, N, ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks),
                               width, height, eps2d, near_plane, far_plane, radius_clip, ptr(radii),
                               ptr(means2d), ptr(depths), ptr(conics), ptr(comp), stream_ptr(dev)),
          "fsgs_project_fwd")
    return radii, means2d, depths, conics, comp


def isect_tiles(means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_width: int,
                tile_height: int, legacy: bool = False, sort: bool = True, return_rule_diff: bool = False):
    """Returns tiles_per_gauss [C,N] i32, isect_ids [M] i64, flatten_ids [M] i32 (sorted if
    ``sort``).  One host sync (n_isects), like the reference.  With ``return_rule_diff`` a fourth
    value says how many Gaussians would be binned differently by the other bbox rule."""
    lib = load()
    dev = means2d.device
    Cn, N = radii.shape
    total = Cn * N
    tpg = torch.empty(Cn, N, dtype=torch.int32, device=dev)
    cum = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
    sbytes = lib.fsgs_scan_scratch_bytes(max(total, 1))
    scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)
    m_host = C.c_int64(0)
    diff_host = C.c_int64(0)
    _run(lib.fsgs_isect_count, (Cn, N, ptr(means2d), ptr(radii), tile_size, tile_width, tile_height,
                               int(legacy), ptr(tpg), ptr(cum), ptr(scratch), sbytes, C.byref(m_host),
                               C.byref(diff_host) if return_rule_diff else None, stream_ptr(dev)),
         "fsgs_isect_count")
    M = int(m_host.value)
    ids = torch.empty(M, dtype=torch.int64, device=dev)
    flat = torch.empty(M, dtype=torch.int32, device=dev)
    if M > 0:
        _run(lib.fsgs_isect_emit, (Cn, N, ptr(means2d), ptr(radii), ptr(depths), ptr(cum), tile_size,
                                  tile_width, tile_height, int(legacy), ptr(ids), ptr(flat),
                                  stream_ptr(dev)), "fsgs_isect_emit")
        if sort:
            n_tiles = tile_width * tile_height
            cam_bits = tile_bits(Cn) if Cn > 1 else 0
            end_bit = 32 + tile_bits(n_tiles) + cam_bits
            ids, flat = sort_pairs(ids, flat, end_bit)
    if return_rule_diff:
        return tpg, ids, flat, int(diff_host.value)
    return tpg, ids, flat


def bin_and_sort(means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_width: int,
                 tile_height: int, legacy: bool = False):
    """isect_tiles + sort + isect_offset_encode in one go.  Returns tiles_per_gauss [C,N] i32,
    isect_ids [M] i64 (sorted), flatten_ids [M] i32, isect_offsets [C,th,tw] i32, rule_diff.
    (count + emit, the stable radix sort, offset_encode)."""
    Cn = radii.shape[0]
    tpg, ids, flat, diff = isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height, legacy=legacy,
                                       sort=True, return_rule_diff=True)
    offsets = isect_offset_encode(ids, Cn, tile_width, tile_height)
    return tpg, ids, flat, offsets, diff


def isect_count_live_async(means2d: Tensor, radii: Tensor, conics: Tensor, opacities: Tensor,
                           tile_width: int, tile_height: int) -> dict:
    """Pass 1 of the live emission WITHOUT the host wait: the total is copied to pinned memory behind an
    event, so the caller can enqueue work that does not need the lists (SH colours, normals, packing)
    before it blocks in ``isect_finish_live``."""
    lib = load()
    dev = means2d.device
    Cn, N = radii.shape
    total = Cn * N
    tpg = torch.empty(Cn, N, dtype=torch.int32, device=dev)
    lpg = torch.empty(Cn, N, dtype=torch.int32, device=dev)
    cum = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
    sbytes = lib.fsgs_scan_scratch_bytes(max(total, 1))
    scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)
    _run(lib.fsgs_isect_count_live, (Cn, N, ptr(means2d), ptr(radii), ptr(conics), ptr(opacities), tile_width,
                                    tile_height, ptr(tpg), ptr(lpg), ptr(cum), ptr(scratch), sbytes, None,
                                    stream_ptr(dev)), "fsgs_isect_count_live")
    st = dict(tpg=tpg, cum=cum, total=total, scratch=scratch, lpg=lpg)
    if total > 0:
        pinned = _pinned_i64(dev)
        pinned.copy_(cum[total - 1:total], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        st.update(pinned=pinned, event=ev)
    return st


_PINNED: dict = {}


SPIN_WAIT = os.environ.get("FSGS_SPIN_WAIT", "1") != "0"
WAIT_ON_FLAG = os.environ.get("FSGS_WAIT_ON_FLAG", "0") == "1"  # (polling the kernel-set flag instead of the event: measured no faster)
# count-free route: the flag is the ONLY signal (no event recorded behind the launch); FSGS_FILL_WAIT_ON_FLAG=0 for A/B
FILL_WAIT_ON_FLAG = os.environ.get("FSGS_FILL_WAIT_ON_FLAG", "1") != "0"


def _wait_event(ev) -> None:
    """The step's one host wait (the live-pair count).  Polling the event instead of sleeping in
    hipEventSynchronize brings the host back some microseconds earlier, and the GPU idles until it is."""
    if SPIN_WAIT:
        for _ in range(200000):
            if ev.query():
                return
    ev.synchronize()


def _pinned_i64(dev) -> Tensor:
    """A small ring of pinned one-element buffers per device (a frame's read-back must not be overwritten
    by the next frame's before the host has looked at it)."""
    ring = _PINNED.setdefault(str(dev), dict(bufs=[torch.zeros(1, dtype=torch.int64).pin_memory() for _ in range(4)], i=0))
    ring["i"] = (ring["i"] + 1) % len(ring["bufs"])
    return ring["bufs"][ring["i"]]


def isect_finish_live(st: dict, means2d: Tensor, radii: Tensor, depths: Tensor, conics: Tensor, opacities: Tensor,
                      tile_width: int, tile_height: int, want_ids: bool = True, route_hint: bool = False):
    """Wait for the live total, then emit + sort + offsets.  Returns what ``bin_and_sort_live`` returns.
    ``route_hint``: the caller is the fused node choosing its binning route from ``bin_live_is_dense`` — the frame's
    density is noted for the next one (an explicit call to this chain must not redirect the fused node)."""
    lib = load()
    dev = means2d.device
    Cn, N = radii.shape
    M = 0
    if st["total"] > 0:
        _wait_event(st["event"])
        M = int(st["pinned"][0])
    ids = torch.empty(M, dtype=torch.int64, device=dev)
    pay = torch.empty(M, dtype=torch.int32, device=dev)
    if M > 0:
        _run(lib.fsgs_isect_emit_live, (Cn, N, ptr(means2d), ptr(radii), ptr(depths), ptr(conics), ptr(opacities),
                                       ptr(st["cum"]), tile_width, tile_height, ptr(ids), ptr(pay), stream_ptr(dev)),
             "fsgs_isect_emit_live")
        n_tiles = tile_width * tile_height
        if route_hint:
            _DENSE_HINT[(str(dev), N, Cn * n_tiles)] = not use_tile_sort(M, Cn * n_tiles)
        if use_tile_sort(M, Cn * n_tiles):
            ids_s, pay_s, offsets = tile_sort(ids, pay, Cn, tile_width, tile_height, want_ids=want_ids)
            return st["tpg"], ids_s, pay_s, offsets
        end_bit = 32 + tile_bits(n_tiles) + (tile_bits(Cn) if Cn > 1 else 0)
        ids, pay = sort_pairs(ids, pay, end_bit)
    offsets = isect_offset_encode(ids, Cn, tile_width, tile_height)
    return st["tpg"], ids, pay, offsets


USE_TILE_SORT = True
TILE_SORT_MAX_MEAN_BUCKET = int(os.environ.get("FSGS_TILE_SORT_MAX_MEAN", "1024"))  # denser scenes keep the global radix sort (many buckets would not fit LDS)


def use_tile_sort(M: int, T: int) -> bool:
    return USE_TILE_SORT and T <= load().fsgs_tile_sort_max_tiles() and M <= TILE_SORT_MAX_MEAN_BUCKET * T


def tile_sort(ids: Tensor, pay: Tensor, n_cameras: int, tile_width: int, tile_height: int, want_ids: bool = True):
    """Sort (cam|tile|depth key, mask<<28|id payload) pairs by key, ties by id, and build isect_offsets:
    one partition by tile + an LDS sort per tile (csrc/tilesort.hip).  Returns (sorted ids or None,
    sorted payload, offsets [C,th,tw])."""
    lib = load()
    dev = ids.device
    M = ids.numel()
    T = n_cameras * tile_width * tile_height
    offsets = torch.empty(T + 1, dtype=torch.int32, device=dev)
    pay_s = torch.empty_like(pay)
    ids_s = torch.empty_like(ids) if want_ids else None
    sbytes = lib.fsgs_tile_sort_scratch_bytes(M, T)
    scratch = WORKSPACE.take(sbytes, dev)
    _run(lib.fsgs_tile_sort, (M, ptr(ids), ptr(pay), n_cameras, tile_width, tile_height, ptr(offsets), ptr(pay_s),
                              ptr(ids_s), ptr(scratch), sbytes, stream_ptr(dev)), "fsgs_tile_sort")
    WORKSPACE.give(scratch)
    return ids_s, pay_s, offsets[:T].view(n_cameras, tile_height, tile_width)


USE_BIN_LIVE = os.environ.get("FSGS_BIN_LIVE", "1") != "0"
# depth-slab split of large buckets: "auto" = where the mean bucket exceeds the LDS tiers, "always", "never"
BIN_SPLIT = os.environ.get("FSGS_BIN_SPLIT", "auto")


def _pinned_i32(dev) -> Tensor:
    """A ring of mapped pinned (total, done-flag) pairs per device; the flag is cleared here, set by the kernel that
    writes the total.  The tensor carries a numpy view (``._np``) for cheap polling."""
    ring = _PINNED.get(str(dev) + ":i32")
    if ring is None:
        bufs = [torch.zeros(4, dtype=torch.int32).pin_memory() for _ in range(4)]  # [total, done, capacity, -]
        for b in bufs:
            b._np = b.numpy()
        ring = _PINNED.setdefault(str(dev) + ":i32", dict(bufs=bufs, i=0))
    ring["i"] = (ring["i"] + 1) % len(ring["bufs"])
    buf = ring["bufs"][ring["i"]]
    if getattr(buf, "_pending", False):
        # the launch this slot was last armed for was never looked at (an exception between a launch and its
        # bin_live_check, a caller that dropped the state): its kernel may still be about to write total and flag, and a
        # bare flag cannot tell whose it is.  Never re-arm under it — drain the device first (a path no training step takes).
        torch.cuda.synchronize(dev)
    buf._np[1] = 0
    buf._np[2] = 0  # no capacity: the offsets are exact
    buf._np[3] = 0  # (count-free binning: "a tile outgrew its bucket")
    buf._pending = True  # (cleared by _wait_total, the only reader)
    return buf


def _wait_total(st: dict) -> int:
    """The step's one host wait: poll the done-flag the offsets kernel sets in mapped host memory right after the
    total (falls back to the event)."""
    arr = st["pinned"]._np
    ev = st.get("event")
    st["pinned"]._pending = False  # (whatever happens below ends with this launch's total read or the stream drained)
    if SPIN_WAIT and (WAIT_ON_FLAG or ev is None):
        for _ in range(400000):
            if arr[1] != 0:
                return int(arr[0])
    if ev is None:  # (count-free route: the flag's kernel was enqueued on the stream that was current AT THE LAUNCH)
        (st.get("stream") or torch.cuda.current_stream(st["pinned_dev"])).synchronize()
        return int(arr[0])
    _wait_event(ev)
    return int(arr[0])


_DENSE_HINT: dict = {}  # (device, N, T) -> the last frame of this shape was too dense for LDS buckets


def bin_live_is_dense(dev, n_gaussians: int, n_tiles_total: int) -> bool:
    """True if the previous frame of this shape fell back to the list chain + radix sort: the caller then starts
    with that chain (whose count pass it needs anyway) instead of a direct count that would be thrown away."""
    return _DENSE_HINT.get((str(dev), n_gaussians, n_tiles_total), False)


def bin_live_count_async(means2d: Tensor, radii: Tensor, conics: Tensor, opacities: Tensor,
                         tile_width: int, tile_height: int, capacity: int = 0) -> dict:
    """Pass 1 of the direct binning (csrc/isect.hip: fsgs_bin_live_count) WITHOUT the host wait: per-tile
    counts of the live pairs, isect_offsets, and the live total on its way to pinned memory behind an event.
    ``capacity`` > 0: the caller will not wait for the total before the second pass either (see
    project_bin_live_count_async): the offsets are clamped to it on the device, bin_live_check looks later."""
    lib = load()
    dev = means2d.device
    Cn, N = radii.shape
    T = Cn * tile_width * tile_height
    tpg = torch.empty(Cn, N, dtype=torch.int32, device=dev)
    offsets = torch.empty(T + 1, dtype=torch.int32, device=dev)
    tbytes = lib.fsgs_bin_live_table_bytes(Cn, N, tile_width, tile_height)
    table = WORKSPACE.take(tbytes, dev)
    pinned = _pinned_i32(dev)  # mapped host memory: the last kernel of the call writes the total there itself
    if capacity > 0:
        pinned._np[2] = int(capacity)
    _run(lib.fsgs_bin_live_count, (Cn, N, ptr(means2d), ptr(radii), ptr(conics), ptr(opacities), tile_width,
                                  tile_height, ptr(tpg), ptr(offsets), ptr(table), tbytes, pinned.data_ptr(),
                                  stream_ptr(dev)), "fsgs_isect_count_live")
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    return dict(direct=True, tpg=tpg, offsets=offsets, table=table, pinned=pinned, event=ev, T=T, capacity=int(capacity))


def project_bin_live_count_async(means: Tensor, quats: Tensor, log_scales: Tensor, opac_logit: Tensor,
                                 binary_threshold, viewmat: Tensor, K: Tensor, width: int, height: int,
                                 tile_width: int, tile_height: int, out: dict, half: Optional[dict] = None,
                                 capacity: int = 0, sh_pack: Optional[tuple] = None, adam=None,
                                 zcut: Optional[Tensor] = None, kept: Optional[Tensor] = None) -> dict:
    """``fsgs_project_fwd_act`` + ``bin_live_count_async`` in one launch chain (one camera): the count pass projects
    its Gaussians itself.  ``out`` holds the projection's output tensors (scales_exp, opac_sig, radii [1,N],
    means2d, depths, conics).  Returns the same state ``bin_live_count_async`` does."""
    lib = load()
    dev = means.device
    N = means.shape[0]
    T = tile_width * tile_height
    tpg = torch.empty(1, N, dtype=torch.int32, device=dev)
    offsets = torch.empty(T + 1, dtype=torch.int32, device=dev)
    tbytes = lib.fsgs_bin_live_table_bytes(1, N, tile_width, tile_height)
    table = WORKSPACE.take(tbytes, dev)
    pinned = _pinned_i32(dev)
    if capacity > 0:
        # the caller will NOT wait for the live total before the second pass: it sizes the lists for `capacity`
        # entries, the offsets kernel clamps to it (fsgs.h: n_live_mapped[2]) and bin_live_finish checks later
        pinned._np[2] = int(capacity)
    if zcut is not None:
        assert zcut.dtype == torch.float32 and zcut.numel() == T and zcut.is_contiguous()
    if sh_pack is not None or adam is not None or zcut is not None or kept is not None:
        # independent work rides in the count pass's launches (fsgs.h): the SH forward + packing in the table scan's,
        # sh_pack = (degree, campos, features_dc, features_rest, c2w, packed, normals_world, zero_cells); a deferred
        # Adam step (`adam`: an _lib.AdamGroups made by adam_groups()) in the count pass's own
        if sh_pack is not None:
            degree, campos, f_dc, f_rest, c2w, packed, normals_world, zero_cells = sh_pack
            tail = (int(degree), ptr(campos), ptr(f_dc), ptr(f_rest), ptr(c2w), ptr(packed), ptr(normals_world),
                    ptr(zero_cells), int(zero_cells.numel()))
        else:
            tail = (0, None, None, None, None, None, None, None, 0)
        # (the tiles longest list first, for the forward's walk: left by the scan launch's last workgroup)
        tile_order = (torch.empty(T, dtype=torch.int32, device=dev)
                      if (sh_pack is not None and TWO_PASS_TILE_ORDER) else None)
        tail = tail + (C.byref(adam) if adam is not None else None, ptr(zcut), ptr(kept), ptr(tile_order), stream_ptr(dev))
        bt = (0 if binary_threshold is None else 1, 0.0 if binary_threshold is None else float(binary_threshold))
        outs = (ptr(out["scales_exp"]), ptr(out["opac_sig"]), ptr(out["radii"]), ptr(out["means2d"]),
                ptr(out["depths"]), ptr(out["conics"]), tile_width, tile_height, ptr(tpg), ptr(offsets), ptr(table),
                tbytes, pinned.data_ptr())
        if half is not None:
            _run(lib.fsgs_project_bin_live_count_sh_pack_h16,
                 (N, ptr(means), ptr(half["quats"]), ptr(half["scales"]), ptr(half["opacities"]), ptr(opac_logit)) + bt
                 + (ptr(viewmat), ptr(K), width, height, 0.3, 0.01, 1e10, 0.0) + outs + tail, "fsgs_isect_count_live")
        else:
            _run(lib.fsgs_project_bin_live_count_sh_pack,
                 (N, ptr(means), ptr(quats), ptr(log_scales), ptr(opac_logit)) + bt
                 + (ptr(viewmat), ptr(K), width, height, 0.3, 0.01, 1e10, 0.0) + outs + tail, "fsgs_isect_count_live")
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        return dict(direct=True, tpg=tpg, offsets=offsets, table=table, pinned=pinned, event=ev, T=T, capacity=capacity,
                    zcut=zcut, tile_order=tile_order)
    if half is not None:  # BASELINE config #5: quats / log-scales / opacity logits read from their half mirrors
        _run(lib.fsgs_project_bin_live_count_h16,
             (N, ptr(means), ptr(half["quats"]), ptr(half["scales"]), ptr(half["opacities"]), ptr(opac_logit),
              0 if binary_threshold is None else 1, 0.0 if binary_threshold is None else float(binary_threshold),
              ptr(viewmat), ptr(K), width, height, 0.3, 0.01, 1e10, 0.0, ptr(out["scales_exp"]), ptr(out["opac_sig"]),
              ptr(out["radii"]), ptr(out["means2d"]), ptr(out["depths"]), ptr(out["conics"]), tile_width, tile_height,
              ptr(tpg), ptr(offsets), ptr(table), tbytes, pinned.data_ptr(), stream_ptr(dev)), "fsgs_isect_count_live")
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        return dict(direct=True, tpg=tpg, offsets=offsets, table=table, pinned=pinned, event=ev, T=T, capacity=capacity)
    _run(lib.fsgs_project_bin_live_count,
         (N, ptr(means), ptr(quats), ptr(log_scales), ptr(opac_logit), 0 if binary_threshold is None else 1,
          0.0 if binary_threshold is None else float(binary_threshold), ptr(viewmat), ptr(K), width, height, 0.3, 0.01,
          1e10, 0.0, ptr(out["scales_exp"]), ptr(out["opac_sig"]), ptr(out["radii"]), ptr(out["means2d"]),
          ptr(out["depths"]), ptr(out["conics"]), tile_width, tile_height, ptr(tpg), ptr(offsets), ptr(table), tbytes,
          pinned.data_ptr(), stream_ptr(dev)), "fsgs_isect_count_live")
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    return dict(direct=True, tpg=tpg, offsets=offsets, table=table, pinned=pinned, event=ev, T=T, capacity=capacity)


# ---- count-free binning (fsgs_project_bin_live_fill_sh_pack): one enumeration of the pairs instead of two ------------
# A view's frame tells how much room every tile's bucket needs in the view's NEXT frame: count * growth + slack words.
# Round 6: 1.25 -> 2.0.  `bench.py --config 2 --steps 300 --densify` (three refinements of +5 % Gaussians each, 100 views, each
# view seen once between two refinements): growth 1.25 / 1.5 / 2.0 -> 90 / 19 / 0 of the 300 frames outgrew a bucket and were
# redone (+0.25 ms each; GPU ms per step over the last three quarters 0.77 / 0.58 / 0.58-0.62) — densification adds its
# Gaussians where the gradients are, so a few tiles grow far more than the model; the room costs 8 bytes of address space per
# spare word and no time (config #2 without refinements: 2048 it/s either way).  FSGS_BIN_FILL_GROWTH: A/B.
BIN_FILL_GROWTH, BIN_FILL_SLACK = float(os.environ.get("FSGS_BIN_FILL_GROWTH", "2.0")), 64
# the count-free route's in-tile sorts are dispatched longest tile first (fsgs.h: tile_order)
BIN_FILL_SORT_ORDER = True
TWO_PASS_TILE_ORDER = True  # (... and the two-pass route's scan launch leaves the same order for the forward's walk)
_BIN_CURSORS: dict = {}  # (device, T) -> int32 [T], zeroed once: the route's own bookkeeping leaves it zeroed


def bin_fill_history_from_offsets(offsets: Tensor, T: int) -> Tensor:
    """bucket bases [T + 1] for the next frame of a view from this frame's EXACT isect_offsets (a frame that went
    through the two-pass route: a view's first visit, a redone frame) — the arithmetic of tile_scan.h:
    bucket_offsets_body256.  A few torch launches, on such frames only."""
    cnt = (offsets[1:T + 1] - offsets[:T]).to(torch.float32)
    room = (cnt * BIN_FILL_GROWTH).to(torch.int32) + BIN_FILL_SLACK
    base = torch.zeros(T + 1, dtype=torch.int32, device=offsets.device)
    base[1:] = torch.cumsum(room, 0)
    return base


def bin_fill_words(n_live_prev: int, T: int) -> int:
    """An upper bound, known on the host, of bucket_base[T] made from a frame of ``n_live_prev`` live pairs."""
    return int(BIN_FILL_GROWTH * n_live_prev) + (BIN_FILL_SLACK + 1) * T + 64


def project_bin_live_fill_async(means: Tensor, quats: Tensor, log_scales: Tensor, opac_logit: Tensor, binary_threshold,
                                viewmat: Tensor, K: Tensor, width: int, height: int, tile_width: int, tile_height: int,
                                out: dict, capacity: int, sh_pack: tuple, hist: dict) -> dict:
    """The count-free route's first call (one camera, fp32, no cuts): projection + ONE enumeration that fills buckets
    sized from ``hist`` (base [T + 1] int32 of the view's previous frame, n_live = that frame's live pairs) + the SH /
    packing launch with the bookkeeping workgroup.  Returns the state bin_live_finish / bin_live_check take."""
    lib = load()
    dev = means.device
    N = means.shape[0]
    T = tile_width * tile_height
    assert capacity > 0 and hist["base"].numel() == T + 1
    tpg = torch.empty(1, N, dtype=torch.int32, device=dev)
    offsets = torch.empty(T + 1, dtype=torch.int32, device=dev)
    next_base = torch.empty(T + 1, dtype=torch.int32, device=dev)
    cursor = _BIN_CURSORS.get((str(dev), T))
    if cursor is None:
        cursor = _BIN_CURSORS[(str(dev), T)] = torch.zeros(T, dtype=torch.int32, device=dev)
    words = bin_fill_words(int(hist["n_live"]), T)
    buckets = WORKSPACE.take(8 * words, dev)
    # (the in-tile sorts' dispatch order, longest tile first: written by the bookkeeping workgroup)
    tile_order = torch.empty(T, dtype=torch.int32, device=dev) if BIN_FILL_SORT_ORDER else None
    pinned = _pinned_i32(dev)
    pinned._np[2] = int(capacity)
    degree, campos, f_dc, f_rest, c2w, packed, normals_world, zero_cells = sh_pack
    bt = (0 if binary_threshold is None else 1, 0.0 if binary_threshold is None else float(binary_threshold))
    _run(lib.fsgs_project_bin_live_fill_sh_pack,
         (N, ptr(means), ptr(quats), ptr(log_scales), ptr(opac_logit)) + bt
         + (ptr(viewmat), ptr(K), width, height, 0.3, 0.01, 1e10, 0.0, ptr(out["scales_exp"]), ptr(out["opac_sig"]),
            ptr(out["radii"]), ptr(out["means2d"]), ptr(out["depths"]), ptr(out["conics"]), tile_width, tile_height,
            ptr(tpg), ptr(offsets), ptr(hist["base"]), ptr(cursor), ptr(buckets), words, ptr(next_base),
            float(BIN_FILL_GROWTH), int(BIN_FILL_SLACK), pinned.data_ptr(), int(degree), ptr(campos), ptr(f_dc),
            ptr(f_rest), ptr(c2w), ptr(packed), ptr(normals_world), ptr(zero_cells), int(zero_cells.numel()),
            ptr(tile_order), stream_ptr(dev)), "fsgs_isect_count_live")
    # (no event behind the launch: the bookkeeping workgroup raises a flag in mapped host memory right after the total,
    # and an event record is a marker packet of its own — ~6 us of idle GPU in front of the next launch, every step)
    ev = None
    if not FILL_WAIT_ON_FLAG:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
    return dict(direct=True, fill=True, tpg=tpg, offsets=offsets, table=None, pinned=pinned, event=ev, T=T, pinned_dev=dev,
                stream=torch.cuda.current_stream(dev),
                capacity=int(capacity), buckets=buckets, base=hist["base"], next_base=next_base, tile_order=tile_order)


class LiveListOverflow(RuntimeError):
    """A frame held more live pairs than the capacity its buffers were sized for (no-wait binning): its lists were
    truncated on the device; ``needed`` is the true total."""

    def __init__(self, needed: int, capacity: int):
        super().__init__(f"live pairs {needed} > capacity {capacity}")
        self.needed, self.capacity = needed, capacity


class OcclusionCutInvalid(RuntimeError):
    """A frame binned with per-tile occlusion cuts (fsgs_tile_zcut_update) in which a cut tile did not saturate inside
    its depth prefix: entries behind the cut could have contributed — the frame must be redone without cuts."""


def zcut_verdict_buffer(dev) -> Tensor:
    """Host-mapped int32 (verdict, landed-flag, -, -) from a ring of its own; the flag is cleared."""
    ring = _PINNED.get(str(dev) + ":zcut")
    if ring is None:
        bufs = [torch.zeros(4, dtype=torch.int32).pin_memory() for _ in range(4)]
        for b in bufs:
            b._np = b.numpy()
        ring = _PINNED.setdefault(str(dev) + ":zcut", dict(bufs=bufs, i=0))
    ring["i"] = (ring["i"] + 1) % len(ring["bufs"])
    buf = ring["bufs"][ring["i"]]
    buf._np[0] = 0
    buf._np[1] = 0
    return buf


from .capacity import LRU as _LRU
_ZCUT_SCRATCH = _LRU(16)  # (one entry per (device, tile count): a handful of image shapes)


def zcut_scratch(dev, T: int):
    """(tile_open [T], bad [1], zcut_cand [T]; int32): zeroed once, left zeroed by fsgs_tile_zcut_update / _recheck."""
    key = (str(dev), T)
    t = _ZCUT_SCRATCH.get(key)
    if t is None:
        t = _ZCUT_SCRATCH[key] = torch.zeros(2 * T + 1, dtype=torch.int32, device=dev)
    return t[:T], t[T:T + 1], t[T + 1:]


def zcut_check(pinned: Tensor) -> int:
    """Waits for a verdict of fsgs_tile_zcut_update / fsgs_tile_zcut_recheck (normally long landed): 0 = the frame is
    exact, 1 = invalid, 2 = open tiles under cuts exist (fsgs_tile_zcut_recheck decides)."""
    arr = pinned._np
    for _ in range(2000000):
        if arr[1] != 0:
            return int(arr[0])
    torch.cuda.synchronize()
    return int(arr[0])


zcut_rechecks = 0  # frames that needed the second look


def zcut_recheck(means2d: Tensor, radii: Tensor, depths: Tensor, conics: Tensor, opacities: Tensor, tile_width: int,
                 tile_height: int, zcut_in: Tensor) -> bool:
    """The second look (fsgs_tile_zcut_recheck) + wait; True = no candidate tile lost a live pair: the frame is exact."""
    global zcut_rechecks
    zcut_rechecks += 1
    lib = load()
    dev = means2d.device
    _, bad, cand = zcut_scratch(dev, tile_width * tile_height)
    verdict = zcut_verdict_buffer(dev)
    _run(lib.fsgs_tile_zcut_recheck, (radii.shape[-1], ptr(means2d), ptr(radii), ptr(depths), ptr(conics), ptr(opacities),
                                     tile_width, tile_height, ptr(zcut_in), ptr(cand), ptr(bad), verdict.data_ptr(),
                                     stream_ptr(dev)), "fsgs_tile_zcut_update")
    return zcut_check(verdict) == 0


def bin_live_check(st: dict) -> int:
    """The deferred look at a no-wait frame's live total (by now it has normally long arrived in mapped memory).
    Returns the total; raises LiveListOverflow if it exceeded the frame's capacity."""
    M = _wait_total(st)
    if st.get("capacity", 0) and M > st["capacity"]:
        raise LiveListOverflow(M, st["capacity"])
    if st.get("fill") and st["pinned"]._np[3] != 0:  # (count-free route: a tile outgrew its bucket — redo, two passes)
        raise LiveListOverflow(M, st["capacity"])
    return M


def bin_live_finish(st: dict, means2d: Tensor, radii: Tensor, depths: Tensor, conics: Tensor, opacities: Tensor,
                    tile_width: int, tile_height: int, long_flag: Optional[Tensor] = None, rel_gate: int = 0):
    """Wait for the live total, then scatter into the tile buckets + sort them.  Returns (tiles_per_gauss,
    None, payload_sorted, isect_offsets [C,th,tw]) — or falls back to the list chain (radix sort) when the
    tiles are too dense for LDS buckets."""
    lib = load()
    dev = means2d.device
    Cn, N = radii.shape
    T = st["T"]
    # no-wait mode: everything below is sized and launched for the capacity; the true total stays on the device
    # (isect_offsets[T], clamped to the capacity) and the caller checks it later with bin_live_check
    M = st["capacity"] if st.get("capacity", 0) else _wait_total(st)
    if st.get("fill"):  # count-free route: the buckets are filled, sort each into the compact payload list
        pay_s = torch.empty(M, dtype=torch.int32, device=dev)
        _run(lib.fsgs_bin_live_sort_buckets, (tile_width, tile_height, ptr(st["offsets"]), ptr(st["base"]),
                                              ptr(st["buckets"]), ptr(pay_s), ptr(long_flag), int(rel_gate),
                                              ptr(st.get("tile_order")), stream_ptr(dev)), "fsgs_tile_sort")
        WORKSPACE.give(st["buckets"])
        return st["tpg"], None, pay_s, st["offsets"]
    dense = not use_tile_sort(M, T)  # mean bucket beyond the LDS tiers
    # (no-wait mode cannot fall back to the list chain, which needs the exact total on the host: dense frames are
    # always split there)
    split = BIN_SPLIT == "always" or (dense and (BIN_SPLIT == "auto" or bool(st.get("capacity", 0))))
    _DENSE_HINT[(str(dev), N, T)] = dense and not split
    if dense and not split:
        WORKSPACE.give(st["table"])
        st2 = isect_count_live_async(means2d, radii, conics, opacities, tile_width, tile_height)
        return isect_finish_live(st2, means2d, radii, depths, conics, opacities, tile_width, tile_height,
                                 want_ids=False, route_hint=True)
    pay_s = torch.empty(M, dtype=torch.int32, device=dev)
    if M > 0 and split:
        # large buckets are split into depth slabs before the LDS sorts (no radix sort, no emission-order lists)
        sbytes = lib.fsgs_bin_live_split_scratch_bytes(Cn, tile_width, tile_height, M)
        arena = WORKSPACE.take(16 * M + sbytes + 512, dev)
        b2 = (8 * M + 255) // 256 * 256
        _run(lib.fsgs_bin_live_emit_split, (Cn, N, ptr(means2d), ptr(radii), ptr(depths), ptr(conics), ptr(opacities),
                                           tile_width, tile_height, ptr(st["offsets"]), ptr(st["table"]), M,
                                           arena.data_ptr(), arena.data_ptr() + b2, arena.data_ptr() + 2 * b2, sbytes,
                                           ptr(pay_s), ptr(st.get("zcut")), stream_ptr(dev)), "fsgs_tile_sort")
        WORKSPACE.give(arena)
    elif M > 0:
        buckets = WORKSPACE.take(8 * M, dev)
        _run(lib.fsgs_bin_live_emit, (Cn, N, ptr(means2d), ptr(radii), ptr(depths), ptr(conics), ptr(opacities),
                                     tile_width, tile_height, ptr(st["offsets"]), ptr(st["table"]), M, ptr(buckets),
                                     ptr(pay_s), ptr(st.get("zcut")), ptr(long_flag), int(rel_gate), stream_ptr(dev)),
             "fsgs_tile_sort")
        WORKSPACE.give(buckets)
    WORKSPACE.give(st["table"])
    if st.get("capacity", 0):
        return st["tpg"], None, pay_s, st["offsets"]  # all T + 1 entries: the lists' end lives in the last one
    return st["tpg"], None, pay_s, st["offsets"][:T].view(Cn, tile_height, tile_width)


def bin_and_sort_live(means2d: Tensor, radii: Tensor, depths: Tensor, conics: Tensor, opacities: Tensor,
                      tile_width: int, tile_height: int):
    """Live emission + sort + offsets (tile 16).  Returns tiles_per_gauss [C,N] (gsplat's count), sorted
    isect_ids [M_live], payload [M_live] (quadrant mask << 28 | flatten id) and isect_offsets [C,th,tw]."""
    st = isect_count_live_async(means2d, radii, conics, opacities, tile_width, tile_height)
    return isect_finish_live(st, means2d, radii, depths, conics, opacities, tile_width, tile_height)


def sort_pairs(keys: Tensor, vals: Tensor, end_bit: int = 64) -> Tuple[Tensor, Tensor]:
    """Stable radix sort of (i64, i32) pairs on key bits [0, end_bit).  Inputs are clobbered."""
    lib = load()
    n = keys.numel()
    if n <= 1:
        return keys, vals
    dev = keys.device
    kb = torch.empty_like(keys)
    vb = torch.empty_like(vals)
    sbytes = lib.fsgs_sort_scratch_bytes(n)
    scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)
    in_b = C.c_int(0)
    _run(lib.fsgs_sort_pairs, (n, ptr(keys), ptr(vals), ptr(kb), ptr(vb), end_bit, ptr(scratch), sbytes,
                              C.byref(in_b), stream_ptr(dev)), "fsgs_sort_pairs")
    return (kb, vb) if in_b.value else (keys, vals)


def isect_offset_encode(isect_ids: Tensor, n_cameras: int, tile_width: int, tile_height: int) -> Tensor:
    lib = load()
    dev = isect_ids.device
    n_tiles = tile_width * tile_height
    offsets = torch.empty(n_cameras, tile_height, tile_width, dtype=torch.int32, device=dev)
    _run(lib.fsgs_isect_offset_encode, (isect_ids.numel(), ptr(isect_ids), n_cameras, n_tiles,
                                       tile_bits(n_tiles), ptr(offsets), stream_ptr(dev)),
          "fsgs_isect_offset_encode")
    return offsets


# --------------------------------------------------------------------------------------
# autograd Functions
# --------------------------------------------------------------------------------------
class _Projection(torch.autograd.Function):
    """fully_fused_projection."""

    @staticmethod
    def forward(ctx, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
                radius_clip, calc_compensations):
        means, quats, scales, viewmats, Ks = map(_c, (means, quats, scales, viewmats, Ks))
        radii, means2d, depths, conics, comp = project_fwd(
            means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip,
            calc_compensations)
        ctx.save_for_backward(means, quats, scales, viewmats, Ks, radii, conics,
                              comp if comp is not None else torch.empty(0, device=means.device))
        ctx.dims = (width, height, eps2d, calc_compensations)
        ctx.mark_non_differentiable(radii)
        if comp is None:
            comp = torch.empty(0, device=means.device)
            ctx.mark_non_differentiable(comp)
        return radii, means2d, depths, conics, comp

    @staticmethod
    def backward(ctx, _v_radii, v_means2d, v_depths, v_conics, v_comp):
        means, quats, scales, viewmats, Ks, radii, conics, comp = ctx.saved_tensors
        width, height, eps2d, calc_comp = ctx.dims
        lib = load()
        dev = means.device
        Cn, N = radii.shape
        v_means2d = _c(v_means2d) if v_means2d is not None else torch.zeros(Cn, N, 2, device=dev)
        v_depths = _c(v_depths) if v_depths is not None else torch.zeros(Cn, N, device=dev)
        v_conics = _c(v_conics) if v_conics is not None else torch.zeros(Cn, N, 3, device=dev)
        use_comp = calc_comp and v_comp is not None
        v_means = torch.empty_like(means)
        v_quats = torch.empty_like(quats)
        v_scales = torch.empty_like(scales)
        v_viewmats = torch.zeros_like(viewmats) if ctx.needs_input_grad[3] else None
        _run(lib.fsgs_project_bwd, (Cn, N, ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks),
                                   width, height, eps2d, ptr(radii), ptr(conics),
                                   ptr(comp) if use_comp else None, ptr(v_means2d), ptr(v_depths),
                                   ptr(v_conics), ptr(_c(v_comp)) if use_comp else None, ptr(v_means),
                                   ptr(v_quats), ptr(v_scales), ptr(v_viewmats), stream_ptr(dev)),
              "fsgs_project_bwd")
        return (v_means, v_quats, v_scales, v_viewmats, None, None, None, None, None, None, None, None)


class _SHColors(torch.autograd.Function):
    """spherical_harmonics fused with the rasterization() glue: dirs, +0.5, clamp_min(0) and the
    optional depth channel."""

    @staticmethod
    def forward(ctx, means, coeffs, campos, radii, depths, degree):
        means, coeffs, campos = map(_c, (means, coeffs, campos))
        depths = _c(depths)
        lib = load()
        dev = means.device
        Cn, N = radii.shape
        K = coeffs.shape[1]
        D = 4 if depths is not None else 3
        colors = torch.empty(Cn, N, D, dtype=torch.float32, device=dev)
        _run(lib.fsgs_sh_fwd, (Cn, N, K, degree, ptr(means), ptr(campos), ptr(coeffs), ptr(radii),
                              ptr(depths), ptr(colors), stream_ptr(dev)), "fsgs_sh_fwd")
        ctx.save_for_backward(means, coeffs, campos, radii)
        ctx.meta = (degree, D, depths is not None)
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        means, coeffs, campos, radii = ctx.saved_tensors
        degree, D, has_depth = ctx.meta
        lib = load()
        dev = means.device
        Cn, N = radii.shape
        K = coeffs.shape[1]
        v_colors = _c(v_colors)
        v_coeffs = torch.empty_like(coeffs)
        v_means = torch.zeros_like(means)
        v_depths = torch.empty(Cn, N, dtype=torch.float32, device=dev) if has_depth else None
        _run(lib.fsgs_sh_bwd, (Cn, N, K, degree, ptr(means), ptr(campos), ptr(coeffs), ptr(radii), D,
                              ptr(v_colors), ptr(v_coeffs), ptr(v_means), ptr(v_depths), stream_ptr(dev)),
              "fsgs_sh_bwd")
        v_campos = None
        if ctx.needs_input_grad[2]:
            # dirs = means - campos: the camera centre receives minus the sum of the per-Gaussian direction gradients
            # (v_means holds exactly the SH share here: it was zero before the launch).  One camera.
            assert Cn == 1, "differentiable camera centre: one camera per call"
            v_campos = -v_means.sum(dim=0, keepdim=True)
        return v_means, v_coeffs, v_campos, None, v_depths, None


USE_LIVE_LISTS = True  # tile_size == 16 fast path (csrc/live.hip); False forces the generic tile kernels


class _Workspace:
    """Size-stable pool for the big per-frame scratch buffers (live-list records are 192 B per
    list entry worst case and their size follows M, which changes every frame: handing such
    requests to the caching allocator makes it split and re-malloc blocks all the time).
    Buffers are handed out with 25 % slack and come back when the consumer is done."""

    def __init__(self):
        self.free = {}

    def take(self, nbytes: int, device) -> Tensor:
        lst = self.free.setdefault(str(device), [])
        best = None
        for i, t in enumerate(lst):
            if t.numel() >= nbytes and (best is None or t.numel() < lst[best].numel()):
                best = i
        if best is not None:
            return lst.pop(best)
        return torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)

    def give(self, t: Optional[Tensor]) -> None:
        if t is not None:
            lst = self.free.setdefault(str(t.device), [])
            lst.append(t)
            if len(lst) > 8:  # drop the smallest
                lst.sort(key=lambda x: x.numel())
                lst.pop(0)


WORKSPACE = _Workspace()


@functools.lru_cache(maxsize=256)
def quad_stream_sizes(Cn: int, tw: int, th: int, M: int):
    """(fsgs_quad_stream_capacity, fsgs_quad_seg_slots) of a frame shape — pure functions of their arguments, asked
    once per shape (with the no-wait binning M is the frame shape's capacity and repeats from frame to frame)."""
    lib = load()
    return int(lib.fsgs_quad_stream_capacity(Cn, tw, th, M)), int(lib.fsgs_quad_seg_slots(Cn, tw, th, M))
_PACKED_ACC: dict = {}


def _packed_accumulator(dev, n_rows: int) -> Tensor:
    """The [n_rows,16] gradient accumulator of the compositing backward on the drop-in route: one per (device, size),
    all zeros between uses (its reader clears it)."""
    key = (str(dev), n_rows)
    t = _PACKED_ACC.get(key)
    if t is None:
        if len(_PACKED_ACC) > 4:
            _PACKED_ACC.clear()
        t = _PACKED_ACC[key] = torch.zeros(n_rows, 16, dtype=torch.float32, device=dev)
    return t


class _Rasterize(torch.autograd.Function):
    """rasterize_to_pixels.  D in {1,3,4}.

    With ``isect_ids`` (the sorted keys) and tile_size 16 the compositing runs on "live lists": every
    list entry gets an exact 4-bit reach mask over the tile's 8x8 quadrants (fsgs_live_payload), and a
    workgroup per quadrant filters, gathers and composites its entries (fsgs_raster_fwd_quad), leaving
    the walked records behind for the segment-parallel backward; otherwise the generic tile kernels
    walk the raw lists.  Both give the same images, last_ids and gradients."""

    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, isect_offsets, flatten_ids, width,
                height, tile_size, absgrad, isect_ids=None, normalize_last=False, live_payload=None,
                ends_on_device=False):
        """``live_payload`` (with ``isect_offsets`` = the LIVE lists' offsets and flatten_ids ignored): the sorted
        live list of the frame as the direct binning leaves it (quadrant mask << 28 | flatten id) — no payload pass.
        ``ends_on_device``: no-wait binning — ``live_payload`` is sized for a CAPACITY, ``isect_offsets`` is a
        [C,th,tw] view of T + 1 stored entries whose last one holds the lists' true end (fsgs.h: n_isects < 0)."""
        means2d_in = means2d
        means2d, conics, colors, opacities = map(_c, (means2d, conics, colors, opacities))
        backgrounds = _c(backgrounds)
        lib = load()
        dev = means2d.device
        Cn, N = opacities.shape
        D = colors.shape[-1]
        th, tw = isect_offsets.shape[1:]
        M = live_payload.numel() if live_payload is not None else flatten_ids.numel()
        render = torch.empty(Cn, height, width, D, dtype=torch.float32, device=dev)
        alphas = torch.empty(Cn, height, width, 1, dtype=torch.float32, device=dev)
        last_ids = torch.empty(Cn, height, width, dtype=torch.int32, device=dev)
        live = USE_LIVE_LISTS and tile_size == 16 and (isect_ids is not None or live_payload is not None)
        if normalize_last and not live:
            raise ValueError("normalize_last is implemented by the live-list kernels only")
        if live:
            a = lambda n: (n + 255) // 256 * 256  # noqa: E731
            needs_bwd = any(ctx.needs_input_grad[:5])
            n_tiles = Cn * tw * th
            packed = torch.empty(Cn * N, 16, dtype=torch.float32, device=dev)
            _run(lib.fsgs_live_pack, (D, Cn * N, ptr(means2d), ptr(conics), ptr(colors), ptr(opacities), None,
                                     ptr(packed), None, 0, stream_ptr(dev)), "fsgs_live_pack", f"_d{D}")
            if live_payload is not None:
                payload = live_payload
            else:
                # quadrant mask << 28 | flatten id for every entry of the caller's sorted lists
                payload = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
                _run(lib.fsgs_live_payload, (ptr(isect_ids), ptr(flatten_ids), M, ptr(packed), Cn * N, tw,
                                            tile_bits(tw * th), ptr(payload), stream_ptr(dev)), "fsgs_live_payload")
            if needs_bwd:
                cap, slots = quad_stream_sizes(Cn, tw, th, M)
                rec_bytes = 4 * cap * 48
                seg_bytes = 4 * slots * 64 * (1 + D) * 4
                nrec_bytes = 4 * n_tiles * 4
                arena = WORKSPACE.take(a(rec_bytes) + a(seg_bytes) + a(nrec_bytes), dev)
                records = arena[:rec_bytes].view(torch.float32)
                seg_state = arena[a(rec_bytes):a(rec_bytes) + seg_bytes].view(torch.float32)
                o = a(rec_bytes) + a(seg_bytes)
                n_rec = arena[o:o + nrec_bytes].view(torch.int32)
            else:
                arena = records = seg_state = n_rec = None
            assert not ends_on_device or live_payload is not None
            _run(lib.fsgs_raster_fwd_quad, (Cn, D, ptr(packed), ptr(payload), isect_offsets.data_ptr(),
                                           -M if ends_on_device else M,
                                           ptr(backgrounds), width, height, tw, th, int(normalize_last),
                                           ptr(render), ptr(alphas), ptr(last_ids), ptr(records), ptr(n_rec),
                                           ptr(seg_state), None, None, None, 0, None, 0, 0, 0, 0, None, 0, None, None, 0, None, stream_ptr(dev)),
                 "fsgs_raster_fwd_quad", f"_d{D}")
            empty = torch.empty(0, device=dev)
            ctx.save_for_backward(records if records is not None else empty,
                                  n_rec if n_rec is not None else empty,
                                  backgrounds if backgrounds is not None else empty,
                                  isect_offsets, alphas, last_ids, render,
                                  seg_state if seg_state is not None else empty)
            # (grad mode is always off inside Function.forward: ask the tape, not torch.is_grad_enabled)
            ctx.arena = arena  # returned to the pool at the end of backward
        else:
            _run(lib.fsgs_raster_fwd, (Cn, N, D, ptr(means2d), ptr(conics), ptr(colors), ptr(opacities),
                                      ptr(backgrounds), width, height, tile_size, tw, th, ptr(isect_offsets),
                                      ptr(flatten_ids), M, ptr(render), ptr(alphas), ptr(last_ids),
                                      stream_ptr(dev)), "fsgs_raster_fwd", f"_d{D}")
            ctx.save_for_backward(means2d, conics, colors, opacities,
                                  backgrounds if backgrounds is not None else torch.empty(0, device=dev),
                                  isect_offsets, flatten_ids, alphas, last_ids)
        ctx.dims = (width, height, tile_size, absgrad, backgrounds is not None, live, Cn, N, D, M,
                    bool(normalize_last))
        # the tensor object the caller holds (meta["means2d"]) receives `.absgrad` in backward
        ctx.means2d_obj = means2d_in
        ctx.mark_non_differentiable(last_ids)
        return render, alphas, last_ids

    @staticmethod
    def backward(ctx, v_render, v_alphas, _v_last):
        width, height, tile_size, absgrad, has_bg, live, Cn, N, D, M, normalize_last = ctx.dims
        lib = load()
        if live:
            records, n_rec, backgrounds, isect_offsets, alphas, last_ids, render, seg_state = ctx.saved_tensors
        else:
            (means2d, conics, colors, opacities, backgrounds, isect_offsets, flatten_ids, alphas,
             last_ids) = ctx.saved_tensors
        dev = alphas.device
        th, tw = isect_offsets.shape[1:]
        v_render = _c(v_render) if v_render is not None else torch.zeros(Cn, height, width, D, device=dev)
        v_alphas = _c(v_alphas) if v_alphas is not None else torch.zeros(Cn, height, width, 1, device=dev)
        if live:
            # persistent, kept zeroed: the unpack launch below clears it again after reading (no 64 B/Gaussian fill)
            v_packed = _packed_accumulator(dev, Cn * N)
            _run(lib.fsgs_raster_bwd_quad, (Cn, D, ptr(records), ptr(n_rec), ptr(isect_offsets), M,
                                           ptr(backgrounds) if has_bg else None, width, height, tw, th,
                                           int(normalize_last), ptr(render), ptr(alphas), ptr(last_ids),
                                           ptr(v_render), ptr(v_alphas),
                                           ptr(seg_state), int(bool(absgrad)), None, None, ptr(v_packed),
                                           DROPIN_BWD_STRIDE, None, None, 0, None, stream_ptr(dev)),
                 "fsgs_raster_bwd_quad", f"_d{D}")
            v_means2d = torch.empty(Cn, N, 2, dtype=torch.float32, device=dev)
            v_conics = torch.empty(Cn, N, 3, dtype=torch.float32, device=dev)
            v_colors = torch.empty(Cn, N, D, dtype=torch.float32, device=dev)
            v_opacities = torch.empty(Cn, N, dtype=torch.float32, device=dev)
            v_abs = torch.empty(Cn, N, 2, dtype=torch.float32, device=dev) if absgrad else None
            _run(lib.fsgs_raster_unpack_grads, (Cn * N, D, ptr(v_packed), 1, ptr(v_means2d), ptr(v_abs),
                                               ptr(v_conics), ptr(v_colors), ptr(v_opacities), None, None,
                                               stream_ptr(dev)), "fsgs_raster_unpack_grads")
        else:
            v_means2d = torch.zeros_like(means2d)
            v_conics = torch.zeros_like(conics)
            v_colors = torch.zeros_like(colors)
            v_opacities = torch.zeros_like(opacities)
            v_abs = torch.zeros_like(means2d) if absgrad else None
            _run(lib.fsgs_raster_bwd, (Cn, N, D, ptr(means2d), ptr(conics), ptr(colors), ptr(opacities),
                                      ptr(backgrounds) if has_bg else None, width, height, tile_size, tw, th,
                                      ptr(isect_offsets), ptr(flatten_ids), M, ptr(alphas), ptr(last_ids),
                                      ptr(v_render), ptr(v_alphas), ptr(v_means2d), ptr(v_abs), ptr(v_conics),
                                      ptr(v_colors), ptr(v_opacities), stream_ptr(dev)), "fsgs_raster_bwd",
                 f"_d{D}")
        if live:
            WORKSPACE.give(getattr(ctx, "arena", None))
            ctx.arena = None
        if absgrad:
            ctx.means2d_obj.absgrad = v_abs
        v_bg = None
        if has_bg and ctx.needs_input_grad[4]:
            v_bg = (v_render * (1.0 - alphas)).sum(dim=(1, 2))
        ctx.means2d_obj = None
        return (v_means2d, v_conics, v_colors, v_opacities, v_bg, None, None, None, None, None, None, None,
                None, None, None)


class _GaussianNormals(torch.autograd.Function):
    """Per-Gaussian normals of dn_splatter/dn_model.py:618-636 in one kernel."""

    @staticmethod
    def forward(ctx, quats, log_scales, means, c2w):
        quats, log_scales, means, c2w = map(_c, (quats, log_scales, means, c2w))
        lib = load()
        dev = quats.device
        N = quats.shape[0]
        n_world = torch.empty(N, 3, dtype=torch.float32, device=dev)
        n_cam = torch.empty(N, 3, dtype=torch.float32, device=dev)
        _run(lib.fsgs_normals_fwd, (N, ptr(quats), ptr(log_scales), ptr(means), ptr(c2w), ptr(n_world),
                                   ptr(n_cam), stream_ptr(dev)), "fsgs_normals_fwd")
        ctx.save_for_backward(quats, log_scales, means, c2w)
        ctx.mark_non_differentiable(n_world)
        return n_world, n_cam

    @staticmethod
    def backward(ctx, _v_world, v_cam):
        quats, log_scales, means, c2w = ctx.saved_tensors
        lib = load()
        dev = quats.device
        N = quats.shape[0]
        v_quats = torch.empty_like(quats)
        _run(lib.fsgs_normals_bwd, (N, ptr(quats), ptr(log_scales), ptr(means), ptr(c2w), ptr(_c(v_cam)),
                                   ptr(v_quats), stream_ptr(dev)), "fsgs_normals_bwd")
        return v_quats, None, None, None


def gaussian_normals(quats: Tensor, log_scales: Tensor, means: Tensor, c2w: Tensor):
    """-> (normals_world [N,3] detached, normals_cam [N,3] differentiable w.r.t. quats)."""
    return _GaussianNormals.apply(quats, log_scales, means.detach(), c2w.detach()[:3, :4])


def densify_stats_(radii: Tensor, absgrad: Tensor, max_hw: int, xys_grad_norm: Tensor,
                   vis_counts: Tensor, max_2Dsize: Tensor) -> None:
    """In-place SplatfactoModel.after_train accumulation."""
    lib = load()
    N = radii.numel()
    _run(lib.fsgs_densify_stats, (N, ptr(radii), ptr(_c(absgrad)), 1.0 / float(max_hw), ptr(xys_grad_norm),
                                 ptr(vis_counts), ptr(max_2Dsize), stream_ptr(radii.device)),
          "fsgs_densify_stats")


def mask_positions(keep: Tensor) -> Tensor:
    """Exclusive scan of a bool/uint8 keep mask -> destination row of every kept row (i64)."""
    lib = load()
    keep8 = keep.to(torch.uint8).contiguous()
    n = keep8.numel()
    pos = torch.empty(max(n, 1), dtype=torch.int64, device=keep.device)
    sbytes = lib.fsgs_scan_scratch_bytes(max(n, 1))
    scratch = torch.empty(sbytes, dtype=torch.uint8, device=keep.device)
    _run(lib.fsgs_mask_scan, (n, ptr(keep8), ptr(pos), ptr(scratch), sbytes, stream_ptr(keep.device)),
          "fsgs_mask_scan")
    return pos[:n]


def compact_rows(src: Tensor, keep8: Tensor, positions: Tensor, n_keep: int,
                 out: Optional[Tensor] = None, out_offset: int = 0) -> Tensor:
    """Order-preserving row compaction of a [n, ...] fp32 tensor.  Writes into
    ``out[out_offset:out_offset+n_keep]`` when given (used to build the grown tensor in place)."""
    lib = load()
    src = src.contiguous()
    n = src.shape[0]
    row = int(src[0].numel()) if n > 0 else 1
    if out is None:
        out = torch.empty((n_keep,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        out_offset = 0
    if n == 0 or n_keep == 0 or src[0].numel() == 0:
        return out  # nothing to move (features_rest is [N,0,3] at sh_degree 0; everything culled; empty source)
    dst = out[out_offset:]
    _run(lib.fsgs_compact_rows, (n, row, ptr(keep8), ptr(positions), ptr(src), dst.data_ptr(),
                                stream_ptr(src.device)), "fsgs_compact_rows")
    return out


def compact_rows_multi(jobs) -> None:
    """``jobs`` = [(src [n, ...] fp32, keep8 [n] uint8, positions [n] int64, dst)]: every job's kept rows of ``src`` land
    at ``dst[positions[row]]`` (dst: a contiguous tensor of the same row shape, possibly a view that starts at a row
    offset), all in ONE launch (fsgs_compact_rows_multi).  Jobs without rows or of zero-width rows are skipped."""
    lib = load()
    keepalive, rows = [], []
    for src, keep8, positions, dst in jobs:
        n = src.shape[0]
        if n == 0 or dst.shape[0] == 0 or src[0].numel() == 0:
            continue
        src = src.contiguous()
        assert src.dtype == torch.float32 and dst.dtype == torch.float32 and dst.is_contiguous()
        keepalive.append(src)
        rows.append((n, int(src[0].numel()), ptr(keep8), ptr(positions), src.data_ptr(), dst.data_ptr()))
    if not rows:
        return
    dev = jobs[0][0].device
    for lo in range(0, len(rows), 32):
        part = rows[lo:lo + 32]
        k = len(part)
        n_rows = (C.c_int64 * k)(*[r[0] for r in part])
        widths = (C.c_int * k)(*[r[1] for r in part])
        arr = lambda col: (C.c_void_p * k)(*[r[col] for r in part])  # noqa: E731
        _run(lib.fsgs_compact_rows_multi, (k, n_rows, widths, arr(2), arr(3), arr(4), arr(5), stream_ptr(dev)),
             "fsgs_compact_rows_multi")


def points_in_boxes(points: Tensor, boxes: Tensor) -> Tensor:
    """[n] bool: point i lies inside any of the oriented boxes [B, 18] (touch.box_frames; fsgs_points_in_boxes)."""
    lib = load()
    points, boxes = points.contiguous(), boxes.contiguous()
    out = torch.empty(points.shape[0], dtype=torch.bool, device=points.device)
    _run(lib.fsgs_points_in_boxes, (points.shape[0], ptr(points), boxes.shape[0], ptr(boxes), ptr(out),
                                    stream_ptr(points.device)), "fsgs_points_in_boxes")
    return out


def nearest_point(queries: Tensor, points: Tensor, want_idx: bool = False, active: Optional[Tensor] = None):
    """Distance from every query [nq,3] to the nearest of points [np,3] (and its index with ``want_idx``).  ``active``
    ([nq] bool / uint8, optional): queries with a zero entry are skipped — distance +inf, index 0."""
    lib = load()
    queries, points = queries.contiguous(), points.contiguous()
    nq, dev = queries.shape[0], queries.device
    dist = torch.empty(nq, dtype=torch.float32, device=dev)
    idx = torch.empty(nq, dtype=torch.int64, device=dev)  # (also the kernel's accumulator: fsgs.h)
    if active is not None:
        active = active.contiguous()
        assert active.numel() == nq and active.element_size() == 1
    _run(lib.fsgs_nearest_point, (nq, ptr(queries), points.shape[0], ptr(points), ptr(active), ptr(dist), ptr(idx),
                                  1 if want_idx else 0, stream_ptr(dev)), "fsgs_nearest_point")
    return (dist, idx) if want_idx else dist


def split_samples(ids: Tensor, n_samples: int, means: Tensor, quats: Tensor, log_scales: Tensor,
                  randn: Tensor) -> Tuple[Tensor, Tensor]:
    lib = load()
    S = ids.numel()
    dev = means.device
    new_means = torch.empty(S * n_samples, 3, dtype=torch.float32, device=dev)
    new_ls = torch.empty(S * n_samples, 3, dtype=torch.float32, device=dev)
    _run(lib.fsgs_split_samples, (S, n_samples, ptr(ids.contiguous()), ptr(_c(means)), ptr(_c(quats)),
                                 ptr(_c(log_scales)), ptr(_c(randn)), ptr(new_means), ptr(new_ls),
                                 stream_ptr(dev)), "fsgs_split_samples")
    return new_means, new_ls


# ---- a-13 / a-14: one refinement / one cull as mark -> [five totals] -> move (csrc/refine.hip) ------------------------------
COL_COPY, COL_ZERO, COL_MEANS, COL_SCALES = 0, 1, 2, 3  # what a NEW row of a column holds (fsgs.h: fsgs_refine_move)
_REFINE_SCRATCH: dict = {}


class RefineMarks:
    """What fsgs_refine_mark left: the flag byte per row, the per-workgroup offsets (device), the five totals (host)."""
    __slots__ = ("n", "flags", "book", "n_keep", "n_split", "n_dup", "n_split_kept", "n_dup_kept", "n_samples")

    def deleted(self) -> Tensor:
        """[n] bool: the rows that do not survive (cull_gaussians' return value)."""
        return (self.flags[:self.n] & 1) == 0


def _u8(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    t = t.contiguous()
    return t.view(torch.uint8) if t.dtype == torch.bool else t


def refine_mark(rules: "_lib.RefineRules", log_scales: Tensor, opac_logit: Tensor, xys_grad_norm: Optional[Tensor] = None,
                vis_counts: Optional[Tensor] = None, max_2Dsize: Optional[Tensor] = None, means: Optional[Tensor] = None,
                add_mask: Optional[Tensor] = None, extra_cull: Optional[Tensor] = None,
                hull_words: Optional[Tensor] = None, boxes: Optional[Tensor] = None, hull_grid=None) -> RefineMarks:
    """fsgs_refine_mark + the one host wait of a refinement: the five totals arrive in mapped memory behind a flag the
    launch's last workgroup raises (no event, no .item())."""
    lib = load()
    dev = log_scales.device
    n = log_scales.shape[0]
    sc = _REFINE_SCRATCH.get(str(dev))
    ints = int(lib.fsgs_refine_book_ints(n))
    if sc is None or sc["flags"].numel() < n or sc["book"].numel() < ints:
        grow = max(n + n // 2, 1024)
        pinned = sc["pinned"] if sc else torch.zeros(8, dtype=torch.int32).pin_memory()
        pinned._np = pinned.numpy()
        sc = _REFINE_SCRATCH[str(dev)] = dict(
            flags=torch.empty(grow, dtype=torch.uint8, device=dev),
            book=torch.zeros(int(lib.fsgs_refine_book_ints(grow)), dtype=torch.int32, device=dev),  # (ticket: zero once)
            pinned=pinned)
    arr = sc["pinned"]._np
    arr[7] = 0
    rules.n_boxes = 0 if boxes is None else int(boxes.shape[0])
    add8, extra8 = _u8(add_mask), _u8(extra_cull)
    boxes = boxes.contiguous() if boxes is not None else None
    g_pts = g_cells = None
    if hull_grid is not None:  # touch.hull_grid: the hull's points by cell — the search rides in the mark
        g_pts, g_cells = hull_grid["points"], hull_grid["cells"]
        for k in range(3):
            rules.grid_origin[k], rules.grid_dims[k] = hull_grid["origin"][k], hull_grid["dims"][k]
        rules.grid_inv_cell = hull_grid["inv_cell"]
    _run(lib.fsgs_refine_mark, (n, C.addressof(rules), ptr(xys_grad_norm), ptr(vis_counts), ptr(max_2Dsize), ptr(means),
                                ptr(_c(log_scales)), ptr(_c(opac_logit)), ptr(add8), ptr(extra8), ptr(hull_words), ptr(g_pts),
                                ptr(g_cells), ptr(boxes), ptr(sc["flags"]), ptr(sc["book"]), sc["pinned"].data_ptr(),
                                stream_ptr(dev)), "fsgs_refine_mark")
    if SPIN_WAIT:
        for _ in range(400000):
            if arr[7] != 0:
                break
    if arr[7] == 0:
        torch.cuda.current_stream(dev).synchronize()
        assert arr[7] != 0, "fsgs_refine_mark ended without its totals"
    m = RefineMarks()
    m.n, m.flags, m.book = n, sc["flags"], sc["book"]
    m.n_keep, m.n_split, m.n_dup, m.n_split_kept, m.n_dup_kept = (int(arr[k]) for k in range(5))
    m.n_samples = int(rules.n_samples) if rules.densify else 0
    return m


def refine_move(marks: RefineMarks, cols, n_append: int = 0, randn: Optional[Tensor] = None, quats: Optional[Tensor] = None,
                log_scales: Optional[Tensor] = None, mask_in: Optional[Tensor] = None, mask_out: Optional[Tensor] = None,
                mask_append: int = 0) -> None:
    """fsgs_refine_move: ``cols`` = [(src [n, ...] fp32, dst, append rows or None, kind)], all in ONE launch."""
    lib = load()
    k = len(cols)
    keep = []
    srcs, dsts, apps, rfs, kinds = [], [], [], [], []
    dev = None
    for src, dst, app, kind in cols:
        assert src.dtype == torch.float32 and dst.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous()
        rf = 1
        for d in src.shape[1:]:
            rf *= int(d)
        if app is not None:
            app = app.to(device=src.device, dtype=torch.float32).contiguous()
            assert app.shape[0] == n_append
            keep.append(app)
        dev = src.device
        srcs.append(src.data_ptr()); dsts.append(dst.data_ptr()); apps.append(app.data_ptr() if app is not None else None)
        rfs.append(rf); kinds.append(int(kind))
    VP = C.c_void_p * max(k, 1)
    IA = C.c_int * max(k, 1)
    if dev is None:
        dev = marks.flags.device
    _run(lib.fsgs_refine_move, (marks.n, k, VP(*srcs), VP(*dsts), VP(*apps), IA(*rfs), IA(*kinds), ptr(marks.flags),
                                ptr(marks.book), marks.n_keep, marks.n_split, marks.n_split_kept, marks.n_dup_kept,
                                int(n_append), marks.n_samples, ptr(randn), ptr(quats), ptr(log_scales), ptr(_u8(mask_in)),
                                ptr(_u8(mask_out)), int(mask_append), stream_ptr(dev)), "fsgs_refine_move")


def refine_keep_all(n: int, dev):
    """(flags, book) of a mark that keeps every one of ``n`` rows and creates none (append_gaussians: no decisions to take,
    so no mark launch): KEEP in every flag byte, workgroup b's first kept row = 256 b."""
    lib = load()
    nb = (n + 255) // 256
    flags = torch.ones(max(n, 1), dtype=torch.uint8, device=dev)
    book = torch.zeros(int(lib.fsgs_refine_book_ints(n)), dtype=torch.int32, device=dev)
    if nb:
        book[8:8 + nb] = torch.arange(nb, dtype=torch.int32, device=dev) * 256
    return flags, book


def nearest_point_words(queries: Tensor, points: Tensor, center, close_radius: float) -> Tensor:
    """[nq] int64 words of fsgs_nearest_point_words: squared-distance bits above index bits of the nearest of ``points``
    for the queries within ``close_radius`` of ``center`` (3 floats), all ones elsewhere."""
    lib = load()
    queries, points = queries.contiguous(), points.contiguous()
    nq, dev = queries.shape[0], queries.device
    words = torch.empty(nq, dtype=torch.int64, device=dev)
    c = (C.c_float * 3)(*[float(x) for x in center])
    _run(lib.fsgs_nearest_point_words, (nq, ptr(queries), points.shape[0], ptr(points), c, float(close_radius), ptr(words),
                                        stream_ptr(dev)), "fsgs_nearest_point_words")
    return words


def loss_combine(partials, weights, bias: float) -> Tensor:
    """bias + sum_t (w[t][0]*colsum0(partials[t]) + w[t][1]*colsum1(partials[t])) as a 0-d device tensor, one launch."""
    lib = load()
    n = len(partials)
    dev = partials[0].device
    out = torch.empty((), dtype=torch.float32, device=dev)
    VP = C.c_void_p * n
    rows = (C.c_int64 * n)(*[p.shape[0] for p in partials])
    w = (C.c_float * (2 * n))(*[float(x) for pair in weights for x in pair])
    _run(lib.fsgs_loss_combine, (n, VP(*[p.data_ptr() for p in partials]), rows, w, float(bias), ptr(out),
                                 stream_ptr(dev)), "fsgs_loss_combine")
    return out


class _SsimL1Loss(torch.autograd.Function):
    """(1-l)*L1 + l*(1-SSIM11) on [H,W,3] images as one forward and one backward kernel (row N2)."""

    @staticmethod
    def forward(ctx, pred, gt, ssim_lambda):
        pred, gt = _c(pred), _c(gt)
        lib = load()
        dev = pred.device
        H, W = pred.shape[0], pred.shape[1]
        maps = torch.empty(3, H, W, 3, dtype=torch.float32, device=dev)
        sums = torch.empty(lib.fsgs_ssim_l1_num_partials(H, W), 2, dtype=torch.float32, device=dev)
        _run(lib.fsgs_ssim_l1_fwd, (H, W, ptr(pred), ptr(gt), maps[0].data_ptr(), maps[1].data_ptr(),
                                    maps[2].data_ptr(), ptr(sums), stream_ptr(dev)), "fsgs_ssim_l1_fwd")
        g_l1 = (1.0 - ssim_lambda) / (3.0 * H * W)
        g_ssim = -ssim_lambda / (3.0 * (H - 10) * (W - 10))
        ctx.save_for_backward(pred, gt, maps)
        ctx.g = (g_l1, g_ssim)
        return loss_combine([sums], [(g_l1, g_ssim)], ssim_lambda)

    @staticmethod
    def backward(ctx, v_loss):
        pred, gt, maps = ctx.saved_tensors
        g_l1, g_ssim = ctx.g
        lib = load()
        dev = pred.device
        H, W = pred.shape[0], pred.shape[1]
        v_pred = torch.empty_like(pred)
        v_loss = v_loss.reshape(1).contiguous().to(torch.float32)
        _run(lib.fsgs_ssim_l1_bwd, (H, W, ptr(pred), ptr(gt), maps[0].data_ptr(), maps[1].data_ptr(),
                                    maps[2].data_ptr(), ptr(v_loss), g_l1, g_ssim, ptr(v_pred),
                                    stream_ptr(dev)), "fsgs_ssim_l1_bwd")
        return v_pred, None, None


def ssim_l1_loss(pred: Tensor, gt: Tensor, ssim_lambda: float = 0.2) -> Tensor:
    return _SsimL1Loss.apply(pred, gt, float(ssim_lambda))


def inverse3x3(A: Tensor) -> Tensor:
    """Differentiable inverse of [..., 3, 3] matrices by cofactors (plain elementwise torch ops: no solver library)."""
    a, b, c = A[..., 0, 0], A[..., 0, 1], A[..., 0, 2]
    d, e, f = A[..., 1, 0], A[..., 1, 1], A[..., 1, 2]
    g, h, i = A[..., 2, 0], A[..., 2, 1], A[..., 2, 2]
    co = torch.stack([torch.stack([e * i - f * h, c * h - b * i, b * f - c * e], -1),
                      torch.stack([f * g - d * i, a * i - c * g, c * d - a * f], -1),
                      torch.stack([d * h - e * g, b * g - a * h, a * e - b * d], -1)], -2)
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    return co / det[..., None, None]


def campos_from_viewmats(viewmats: Tensor) -> Tensor:
    """torch.inverse(viewmats)[:, :3, 3] without a solver library or a host sync."""
    lib = load()
    viewmats = viewmats.detach().contiguous()
    Cn = viewmats.shape[0]
    campos = torch.empty(Cn, 3, dtype=torch.float32, device=viewmats.device)
    _run(lib.fsgs_campos_from_viewmats, (Cn, ptr(viewmats), ptr(campos), stream_ptr(viewmats.device)),
         "fsgs_campos_from_viewmats")
    return campos


class _AuxL1Loss(torch.autograd.Function):
    """w_d * mean|depth - gt| + w_n * mean|normal - gt| as one forward and one backward kernel."""

    @staticmethod
    def forward(ctx, depth, depth_gt, normal, normal_gt, w_depth, w_normal):
        depth, depth_gt, normal, normal_gt = map(_c, (depth, depth_gt, normal, normal_gt))
        lib = load()
        dev = depth.device
        P = depth.numel()
        nblk = (P + 255) // 256
        partial = torch.empty(nblk, 2, dtype=torch.float32, device=dev)
        _run(lib.fsgs_aux_l1_fwd, (P, ptr(depth), ptr(depth_gt), ptr(normal), ptr(normal_gt), ptr(partial),
                                   stream_ptr(dev)), "fsgs_aux_l1_fwd")
        ctx.save_for_backward(depth, depth_gt, normal if normal is not None else torch.empty(0, device=dev),
                              normal_gt if normal_gt is not None else torch.empty(0, device=dev))
        g_d = w_depth / P
        g_n = w_normal / (3.0 * P) if normal is not None else 0.0
        ctx.g = (g_d, g_n, normal is not None)
        return loss_combine([partial], [(g_d, g_n)], 0.0)

    @staticmethod
    def backward(ctx, v_loss):
        depth, depth_gt, normal, normal_gt = ctx.saved_tensors
        g_d, g_n, has_n = ctx.g
        lib = load()
        dev = depth.device
        P = depth.numel()
        v_depth = torch.empty_like(depth)
        v_normal = torch.empty_like(normal) if has_n else None
        v_loss = v_loss.reshape(1).contiguous().to(torch.float32)
        _run(lib.fsgs_aux_l1_bwd, (P, ptr(depth), ptr(depth_gt), ptr(normal) if has_n else None,
                                   ptr(normal_gt) if has_n else None, ptr(v_loss), g_d, g_n, ptr(v_depth),
                                   ptr(v_normal), stream_ptr(dev)), "fsgs_aux_l1_bwd")
        return v_depth, None, v_normal, None, None, None


def aux_l1_loss(depth: Tensor, depth_gt: Tensor, normal: Optional[Tensor], normal_gt: Optional[Tensor],
                w_depth: float, w_normal: float) -> Tensor:
    return _AuxL1Loss.apply(depth, depth_gt, normal, normal_gt, float(w_depth), float(w_normal))


DEFER_COMBINE = os.environ.get("FSGS_DEFER_COMBINE", "1") != "0"


class _TrainLoss(torch.autograd.Function):
    """The whole config-#2 training loss as ONE autograd node:
    (1-l)*L1(rgb) + l*(1-SSIM11(rgb)) + w_d*L1(depth) + w_n*L1(normal)  — two partial-sum kernels and
    one combine launch forward, two kernels backward, no torch glue in between."""

    @staticmethod
    def forward(ctx, rgb, rgb_gt, depth, depth_gt, normal, normal_gt, ssim_lambda, w_depth, w_normal, seed=None,
                aux_done=None):
        """seed: the upstream gradient of the loss (1-element device tensor) when the caller knows it already
        (the trainer's tape-free step): the depth / normal gradient images are then written by the forward's
        own pass over those images and ``backward`` skips its aux launch."""
        rgb, rgb_gt, depth, depth_gt, normal, normal_gt = map(_c, (rgb, rgb_gt, depth, depth_gt, normal, normal_gt))
        lib = load()
        dev = rgb.device
        H, W = rgb.shape[0], rgb.shape[1]
        P = depth.numel()
        maps = torch.empty(3, H, W, 3, dtype=torch.float32, device=dev)
        sums = torch.empty(lib.fsgs_ssim_l1_num_partials(H, W), 2, dtype=torch.float32, device=dev)
        _run(lib.fsgs_ssim_l1_fwd, (H, W, ptr(rgb), ptr(rgb_gt), maps[0].data_ptr(), maps[1].data_ptr(),
                                    maps[2].data_ptr(), ptr(sums), stream_ptr(dev)), "fsgs_ssim_l1_fwd")
        partial = None if aux_done is not None else torch.empty((P + 255) // 256, 2, dtype=torch.float32, device=dev)
        g_l1 = (1.0 - ssim_lambda) / (3.0 * H * W)
        g_ssim = -ssim_lambda / (3.0 * (H - 10) * (W - 10))
        g_d = w_depth / P
        g_n = w_normal / (3.0 * P) if normal is not None else 0.0
        ctx.aux_grads = None
        if aux_done is not None and seed is not None:
            # the depth / normal terms were evaluated with the images (fsgs_epilogue_loss_fwd, same weights and seed)
            partial, v_depth, v_normal = aux_done
            ctx.aux_grads = (seed, v_depth, v_normal)
        elif seed is not None:
            v_depth = torch.empty_like(depth)
            v_normal = torch.empty_like(normal) if normal is not None else None
            _run(lib.fsgs_aux_l1_fwd_bwd, (P, ptr(depth), ptr(depth_gt), ptr(normal), ptr(normal_gt), ptr(partial),
                                           ptr(seed), g_d, g_n, ptr(v_depth), ptr(v_normal), stream_ptr(dev)),
                 "fsgs_aux_l1_fwd")
            ctx.aux_grads = (seed, v_depth, v_normal)
        else:
            _run(lib.fsgs_aux_l1_fwd, (P, ptr(depth), ptr(depth_gt), ptr(normal), ptr(normal_gt), ptr(partial),
                                       stream_ptr(dev)), "fsgs_aux_l1_fwd")
        ctx.save_for_backward(rgb, rgb_gt, maps, depth, depth_gt,
                              normal if normal is not None else torch.empty(0, device=dev),
                              normal_gt if normal_gt is not None else torch.empty(0, device=dev))
        ctx.g = (g_l1, g_ssim, g_d, g_n, normal is not None)
        ctx.deferred_loss = None
        if seed is not None and DEFER_COMBINE:
            # tape-free step: the scalar is only reported, its combine rides in the backward's SSIM launch
            out = torch.empty((), dtype=torch.float32, device=dev)
            ctx.deferred_loss = (out, sums, partial, ssim_lambda)
            return out
        return loss_combine([sums, partial], [(g_l1, g_ssim), (g_d, g_n)], ssim_lambda)

    @staticmethod
    def backward(ctx, v_loss):
        rgb, rgb_gt, maps, depth, depth_gt, normal, normal_gt = ctx.saved_tensors
        g_l1, g_ssim, g_d, g_n, has_n = ctx.g
        lib = load()
        dev = rgb.device
        H, W = rgb.shape[0], rgb.shape[1]
        P = depth.numel()
        v_loss_in = v_loss
        v_loss = v_loss.reshape(1).contiguous().to(torch.float32)
        v_rgb = torch.empty_like(rgb)
        aux = getattr(ctx, "aux_grads", None)
        if aux is not None and aux[0] is v_loss_in:
            v_depth, v_normal = aux[1], aux[2]  # written by the forward's pass with this very seed
        else:
            v_depth = torch.empty_like(depth)
            v_normal = torch.empty_like(normal) if has_n else None
            _run(lib.fsgs_aux_l1_bwd, (P, ptr(depth), ptr(depth_gt), ptr(normal) if has_n else None,
                                       ptr(normal_gt) if has_n else None, ptr(v_loss), g_d, g_n, ptr(v_depth),
                                       ptr(v_normal), stream_ptr(dev)), "fsgs_aux_l1_bwd")
        dl = getattr(ctx, "deferred_loss", None)
        if dl is not None:
            out, sums, partial, ssim_lambda = dl
            ctx.deferred_loss = None
            VP = C.c_void_p * 2
            rows = (C.c_int64 * 2)(sums.shape[0], partial.shape[0])
            w = (C.c_float * 4)(g_l1, g_ssim, g_d, g_n)
            _run(lib.fsgs_ssim_l1_bwd_combine, (H, W, ptr(rgb), ptr(rgb_gt), maps[0].data_ptr(), maps[1].data_ptr(),
                                                maps[2].data_ptr(), ptr(v_loss), g_l1, g_ssim, ptr(v_rgb), 2,
                                                VP(sums.data_ptr(), partial.data_ptr()), rows, w, float(ssim_lambda),
                                                ptr(out), stream_ptr(dev)), "fsgs_ssim_l1_bwd")
        else:
            _run(lib.fsgs_ssim_l1_bwd, (H, W, ptr(rgb), ptr(rgb_gt), maps[0].data_ptr(), maps[1].data_ptr(),
                                        maps[2].data_ptr(), ptr(v_loss), g_l1, g_ssim, ptr(v_rgb),
                                        stream_ptr(dev)), "fsgs_ssim_l1_bwd")
        return v_rgb, None, v_depth, None, v_normal, None, None, None, None, None, None


def train_loss(rgb: Tensor, rgb_gt: Tensor, depth: Tensor, depth_gt: Tensor, normal: Optional[Tensor],
               normal_gt: Optional[Tensor], ssim_lambda: float = 0.2, w_depth: float = 0.2,
               w_normal: float = 0.1) -> Tensor:
    return _TrainLoss.apply(rgb, rgb_gt, depth, depth_gt, normal, normal_gt, float(ssim_lambda),
                            float(w_depth), float(w_normal), None, None)


def loss_combine_cols(partials, weights, bias: float) -> Tensor:
    """bias + sum_t sum_c w[t][c] * colsum_c(partials[t]) as a 0-d device tensor; partials[t] is [rows, cols]."""
    lib = load()
    n = len(partials)
    dev = partials[0].device
    out = torch.empty((), dtype=torch.float32, device=dev)
    a = _combine_args(partials, weights)
    _run(lib.fsgs_loss_combine_cols, (n, a[0], a[1], a[2], a[3], float(bias), ptr(out), stream_ptr(dev)),
         "fsgs_loss_combine")
    return out


def _combine_args(partials, weights):
    n = len(partials)
    VP = C.c_void_p * n
    rows = (C.c_int64 * n)(*[p.shape[0] for p in partials])
    cols = (C.c_int * n)(*[p.shape[1] for p in partials])
    flat = [float(x) for w, p in zip(weights, partials) for x in list(w)[:p.shape[1]]]
    assert len(flat) == sum(p.shape[1] for p in partials), "one weight per column"
    return VP(*[p.data_ptr() for p in partials]), rows, cols, (C.c_float * len(flat))(*flat)


def depth_valid_counts(sensor_depth: Tensor, mask: Optional[Tensor], depth_tol: float) -> Tuple[int, int, int]:
    """(#valid pixels with a right neighbour, #valid with a lower neighbour, #valid), valid = sensor*mask > tol: the
    sizes of EdgeAwareLogL1's two masked selections (dn_splatter/losses.py:208-209) and of the plain depth-loss
    types' selection (dn_model.py:733).  One host read-back — per VIEW, at batch preparation, never inside a step."""
    lib = load()
    H, W = sensor_depth.shape[0], sensor_depth.shape[1]
    counts = torch.empty(3, dtype=torch.int64, device=sensor_depth.device)
    _run(lib.fsgs_depth_valid_counts, (H, W, ptr(sensor_depth), ptr(mask), float(depth_tol), ptr(counts),
                                       stream_ptr(sensor_depth.device)), "fsgs_depth_valid_counts")
    cx, cy, ca = counts.tolist()
    return int(cx), int(cy), int(ca)


_FA_TYPE = {"EdgeAwareLogL1": 0, "L1": 4, "LogL1": 8, "MSE": 12}  # fsgs.h: FSGS_FA_DEPTH_*


def fusion_aux_flags(cfg) -> int:
    """The flags of fsgs_fusion_aux_loss for a LossConfig (fsgs.h: FSGS_FA_*)."""
    f = _FA_TYPE[getattr(cfg, "depth_loss_type", "EdgeAwareLogL1")]
    if getattr(cfg, "smooth_loss_type", "TV") == "EdgeAwareTV":
        f |= 2
    if getattr(cfg, "normal_supervision", "mono") == "depth":
        f |= 1
    return f


def mono_depth_weights(cfg, fb):
    """The two EdgeAwareLogL1 column weights of the monocular-depth term (dn_model.py:737-745) as a w_aux[7] row, or
    None when the batch carries no mono depth / the term is off."""
    if not (cfg.use_depth_loss and getattr(fb, "mono_depth", None) is not None
            and getattr(cfg, "mono_depth_lambda", 0.0) > 0.0):
        return None
    w = [0.0] * 7
    if getattr(cfg, "depth_loss_type", "EdgeAwareLogL1") == "EdgeAwareLogL1":
        w[0] = cfg.mono_depth_lambda / fb.mono_cnt_x if fb.mono_cnt_x else float("inf")
        w[1] = cfg.mono_depth_lambda / fb.mono_cnt_y if fb.mono_cnt_y else float("inf")
    else:  # (L1 / LogL1 / MSE: one column, mean over the valid pixels)
        w[0] = cfg.mono_depth_lambda / fb.mono_cnt if fb.mono_cnt else float("inf")
    return w


def fusion_loss_weights(cfg, fb, n_gauss: int, n_touch: int):
    """Every partial-sum column's weight in get_loss_dict's total (dn_model.py:673-925), as host floats:
    (g_l1, g_ssim), w_aux[7], g_minscale, g_touch."""
    H, W = fb.height, fb.width
    g_l1 = (1.0 - cfg.ssim_lambda) / (3.0 * H * W)
    g_ssim = -cfg.ssim_lambda / (3.0 * (H - 10) * (W - 10))
    w = [0.0] * 7
    if cfg.use_depth_loss and fb.has_sensor and cfg.sensor_depth_lambda > 0.0:
        # (an empty selection makes the reference's mean NaN; so does 0 * inf here)
        if getattr(cfg, "depth_loss_type", "EdgeAwareLogL1") == "EdgeAwareLogL1":
            w[0] = cfg.sensor_depth_lambda / fb.cnt_x if fb.cnt_x else float("inf")
            w[1] = cfg.sensor_depth_lambda / fb.cnt_y if fb.cnt_y else float("inf")
        else:  # L1 / LogL1 / MSE: column 0 only, mean over the valid pixels (dn_model.py:731-736)
            w[0] = cfg.sensor_depth_lambda / fb.cnt if fb.cnt else float("inf")
    if cfg.use_depth_smooth_loss:  # (TV and EdgeAwareTV average over the same two shifted grids)
        w[2] = cfg.smooth_loss_lambda / (H * (W - 1))
        w[3] = cfg.smooth_loss_lambda / ((H - 1) * W)
    if cfg.use_normal_loss:
        if fb.normal is not None or getattr(cfg, "normal_supervision", "mono") == "depth":
            w[4] = cfg.normal_lambda / (3.0 * H * W)
        if cfg.use_normal_tv_loss:
            w[5] = cfg.normal_lambda / (3.0 * H * (W - 1))
            w[6] = cfg.normal_lambda / (3.0 * (H - 1) * W)
    g_min = cfg.normal_lambda / n_gauss if (cfg.two_d_gaussians and n_gauss > 0) else 0.0
    g_touch = cfg.touch_normal_loss_lambda / (3.0 * n_touch) if n_touch > 0 else 0.0
    return (g_l1, g_ssim), w, g_min, g_touch


def min_scale_grad_(log_scales: Tensor, g: float, seed: Tensor, g_log_scales: Tensor) -> None:
    """g_log_scales[n, argmin] += seed * g * exp(min_k log_scales[n,k]): the gradient of the two_d_gaussians term
    (dn_model.py:817-819), accumulated onto whatever the rasterizer's backward has written."""
    lib = load()
    N = log_scales.shape[0]
    dev = log_scales.device
    scratch = torch.empty((N + 255) // 256, 2, dtype=torch.float32, device=dev)
    _run(lib.fsgs_min_scale_loss, (N, ptr(log_scales), float(g), ptr(seed), ptr(scratch), ptr(g_log_scales),
                                   stream_ptr(dev)), "fsgs_min_scale_loss")


class _FusionLoss(torch.autograd.Function):
    """DNSplatterModel.get_loss_dict (dn_splatter/dn_model.py:673-925; sensor-depth + mono-normal configuration)
    as ONE autograd node: masked L1 + SSIM11 on rgb, EdgeAwareLogL1 + TV on depth, L1 + TV on the normals, the
    min-scale term and the touch-normal MSE — four partial-sum kernels and one combine forward (the depth / normal
    gradient images come out of the forward's own pass when the caller hands in the loss seed), one or two kernels
    backward.  Returns main_loss + scale_reg (scale_reg = 0: use_scale_regularization is off, dn_model.py:120)."""

    @staticmethod
    def forward(ctx, rgb, depth, normal, log_scales, fb, cfg, normals_world=None, touch_idx=None,
                touch_normals=None, seed=None, defer_combine=False, intrinsics=None):
        """``intrinsics`` = (fx, fy, cx, cy) of the frame's camera: read by normal_supervision == "depth" only
        (the reference takes them from ``self.camera``, dn_model.py:775-786)."""
        rgb, depth, normal, log_scales = map(_c, (rgb, depth, normal, log_scales))
        lib = load()
        dev = rgb.device
        sp = stream_ptr(dev)
        H, W = fb.height, fb.width
        N = log_scales.shape[0]
        n_touch = 0 if touch_idx is None else int(touch_idx.numel())
        if n_touch and (touch_idx.dtype != torch.int64 or not touch_idx.is_cuda):
            raise ValueError("touch_idx must be an int64 tensor of row indices on the GPU (fsgs_touch_normal_sqerr)")
        if n_touch and (normals_world is None or touch_normals is None or touch_normals.shape[0] != n_touch):
            raise ValueError("touch anchors need normals_world [N,3] and touch_normals [n_touch,3]")
        (g_l1, g_ssim), w_aux, g_min, g_touch = fusion_loss_weights(cfg, fb, N, n_touch)
        f32 = dict(dtype=torch.float32, device=dev)
        maps = torch.empty(3, H, W, 3, **f32)
        sums = torch.empty(lib.fsgs_ssim_l1_num_partials(H, W), 2, **f32)
        _run(lib.fsgs_ssim_l1_fwd_masked, (H, W, ptr(rgb), ptr(fb.image), ptr(fb.mask), maps[0].data_ptr(),
                                           maps[1].data_ptr(), maps[2].data_ptr(), ptr(sums), sp), "fsgs_ssim_l1_fwd")
        aux = torch.empty(lib.fsgs_fusion_aux_num_partials(H, W), 8, **f32)
        wa = (C.c_float * 7)(*w_aux)
        flags = fusion_aux_flags(cfg)
        edge = (flags & 12) == 0
        # the target of the normal L1 (and cosine) term: the view's monocular normals — or, normal_supervision ==
        # "depth", the pseudo normals of THIS frame's masked depth (detached: no gradient flows through the target)
        normal_gt = fb.normal if normal is not None else None
        if normal is not None and cfg.use_normal_loss and getattr(cfg, "normal_supervision", "mono") == "depth":
            if intrinsics is None:
                raise ValueError("normal_supervision='depth' needs the frame's camera intrinsics (fx, fy, cx, cy)")
            normal_gt = torch.empty(H, W, 3, **f32)
            _run(lib.fsgs_normals_from_depth, (H, W, ptr(depth), ptr(fb.mask), float(intrinsics[0]), float(intrinsics[1]),
                                               float(intrinsics[2]), float(intrinsics[3]), ptr(normal_gt), sp),
                 "fsgs_normals_from_depth")
        elif normal_gt is None:
            flags &= ~1
        v_depth = v_normal = None
        if seed is not None:
            v_depth = torch.empty_like(depth)
            v_normal = torch.empty_like(normal) if normal is not None else None
        # the min-scale term's and the touch-normal term's partial sums ride in this launch (extra workgroups: fsgs.h)
        pm = torch.empty((N + 255) // 256, 2, **f32) if g_min != 0.0 else None
        pt = torch.empty((n_touch + 255) // 256, 2, **f32) if n_touch > 0 else None
        _run(lib.fsgs_fusion_aux_loss_riders, (H, W, ptr(depth), ptr(normal), ptr(fb.image), ptr(fb.sensor_depth),
                                               ptr(normal_gt), ptr(fb.mask),
                                               float(cfg.depth_tolerance), wa, ptr(seed), ptr(aux), ptr(v_depth),
                                               ptr(v_normal), flags,
                                               N if pm is not None else 0, ptr(log_scales), ptr(pm),
                                               n_touch, ptr(touch_idx), ptr(normals_world),
                                               ptr(_c(touch_normals)) if n_touch > 0 else None, ptr(pt), sp),
             "fsgs_fusion_aux_loss")
        partials, weights = [sums, aux], [(g_l1, g_ssim), tuple(w_aux) + (0.0,)]
        w_mono = mono_depth_weights(cfg, fb)
        if w_mono is not None:
            # monocular depth (dn_model.py:737-750): the same depth columns against a second target — a second pass of
            # the kernel with every other column's weight zero; its depth gradient is added.  (The plain depth-loss
            # types take mono > 0 as valid, EdgeAwareLogL1 mono > depth_tolerance: :739-741.)
            aux_m = torch.empty_like(aux)
            v_depth_m = torch.empty_like(depth) if seed is not None else None
            _run(lib.fsgs_fusion_aux_loss, (H, W, ptr(depth), None, ptr(fb.image), ptr(fb.mono_depth), None, ptr(fb.mask),
                                            float(cfg.depth_tolerance) if edge else 0.0, (C.c_float * 7)(*w_mono),
                                            ptr(seed), ptr(aux_m), ptr(v_depth_m), None, flags & 12, sp),
                 "fsgs_fusion_aux_loss")
            if v_depth_m is not None:
                v_depth.add_(v_depth_m)
            partials.append(aux_m)
            weights.append(tuple(w_mono) + (0.0,))
        g_cos = 0.0
        if normal is not None and cfg.use_normal_loss and getattr(cfg, "use_normal_cosine_loss", False) \
                and normal_gt is not None:
            # the cosine term as the reference calls it (dn_model.py:807-813): one value per (channel, column)
            g_cos = cfg.normal_lambda / (3.0 * W)
            pc = torch.empty(lib.fsgs_normal_cosine_num_partials(W), 2, **f32)
            _run(lib.fsgs_normal_cosine_loss, (H, W, ptr(normal), ptr(normal_gt), ptr(fb.mask), flags & 1, g_cos, ptr(seed),
                                               ptr(pc), ptr(v_normal), sp), "fsgs_normal_cosine_loss")
            partials.append(pc)
            weights.append((g_cos, 0.0))
        if pm is not None:
            partials.append(pm)
            weights.append((g_min, 0.0))
        if pt is not None:
            partials.append(pt)
            weights.append((g_touch, 0.0))
        ctx.save_for_backward(rgb, maps, depth, normal if normal is not None else torch.empty(0, device=dev),
                              log_scales)
        ctx.fb, ctx.cfg = fb, cfg
        ctx.g = (g_l1, g_ssim, w_aux, g_min, normal is not None)
        ctx.extra = (normal_gt, flags, g_cos)  # (the frame's own target normals, when they came from its depth)
        ctx.aux_grads = (seed, v_depth, v_normal) if seed is not None else None
        ctx.deferred_loss = None
        if seed is not None and defer_combine:
            # tape-free step: the scalar is only reported, its combine rides in the backward's SSIM launch
            out = torch.empty((), **f32)
            ctx.deferred_loss = (out, partials, weights, cfg.ssim_lambda)
            return out
        return loss_combine_cols(partials, weights, cfg.ssim_lambda)

    @staticmethod
    def backward(ctx, v_loss):
        rgb, maps, depth, normal, log_scales = ctx.saved_tensors
        g_l1, g_ssim, w_aux, g_min, has_n = ctx.g
        fb, cfg = ctx.fb, ctx.cfg
        lib = load()
        dev = rgb.device
        sp = stream_ptr(dev)
        H, W = fb.height, fb.width
        v_loss_in = v_loss
        v_loss = v_loss.reshape(1).contiguous().to(torch.float32)
        aux = ctx.aux_grads
        if aux is not None and aux[0] is v_loss_in:
            v_depth, v_normal = aux[1], aux[2]  # written by the forward's pass with this very seed
        else:
            v_depth = torch.empty_like(depth)
            v_normal = torch.empty_like(normal) if has_n else None
            normal_gt, flags, g_cos = ctx.extra
            edge = (flags & 12) == 0
            scratch = torch.empty(lib.fsgs_fusion_aux_num_partials(H, W), 8, dtype=torch.float32, device=dev)
            _run(lib.fsgs_fusion_aux_loss, (H, W, ptr(depth), ptr(normal) if has_n else None, ptr(fb.image),
                                            ptr(fb.sensor_depth), ptr(normal_gt) if has_n else None, ptr(fb.mask),
                                            float(cfg.depth_tolerance), (C.c_float * 7)(*w_aux), ptr(v_loss),
                                            ptr(scratch), ptr(v_depth), ptr(v_normal), flags, sp), "fsgs_fusion_aux_loss")
            w_mono = mono_depth_weights(cfg, fb)
            if w_mono is not None:
                v_depth_m = torch.empty_like(depth)
                _run(lib.fsgs_fusion_aux_loss, (H, W, ptr(depth), None, ptr(fb.image), ptr(fb.mono_depth), None,
                                                ptr(fb.mask), float(cfg.depth_tolerance) if edge else 0.0,
                                                (C.c_float * 7)(*w_mono), ptr(v_loss), ptr(scratch), ptr(v_depth_m), None,
                                                flags & 12, sp), "fsgs_fusion_aux_loss")
                v_depth.add_(v_depth_m)
            if g_cos != 0.0:
                pc = torch.empty(lib.fsgs_normal_cosine_num_partials(W), 2, dtype=torch.float32, device=dev)
                _run(lib.fsgs_normal_cosine_loss, (H, W, ptr(normal), ptr(normal_gt), ptr(fb.mask), flags & 1, g_cos,
                                                   ptr(v_loss), ptr(pc), ptr(v_normal), sp), "fsgs_normal_cosine_loss")
        v_rgb = torch.empty_like(rgb)
        dl = ctx.deferred_loss
        if dl is not None:
            out, partials, weights, bias = dl
            ctx.deferred_loss = None
            a = _combine_args(partials, weights)
            comb = (len(partials), a[0], a[1], a[2], a[3], float(bias), ptr(out))
        else:
            comb = (0, None, None, None, None, 0.0, None)
        _run(lib.fsgs_ssim_l1_bwd_masked, (H, W, ptr(rgb), ptr(fb.image), ptr(fb.mask), maps[0].data_ptr(),
                                           maps[1].data_ptr(), maps[2].data_ptr(), ptr(v_loss), g_l1, g_ssim,
                                           ptr(v_rgb)) + comb + (sp,), "fsgs_ssim_l1_bwd")
        v_scales = None
        if ctx.needs_input_grad[3]:
            v_scales = torch.zeros_like(log_scales)
            if g_min != 0.0:
                min_scale_grad_(log_scales, g_min, v_loss, v_scales)
        return v_rgb, v_depth, v_normal, v_scales, None, None, None, None, None, None, None, None


def fusion_loss(out, fb, cfg, log_scales: Tensor, touch_idx: Optional[Tensor] = None,
                touch_normals: Optional[Tensor] = None, intrinsics=None) -> Tensor:
    """get_loss_dict's main_loss + scale_reg for the outputs of get_outputs (``out``: rgb, depth, normal,
    normals_world) against a prepared :class:`fusionsense_amd.losses.FrameBatch`; ``intrinsics`` = (fx, fy, cx, cy) of
    the frame's camera (needed by normal_supervision == "depth" only)."""
    return _FusionLoss.apply(out["rgb"], out["depth"], out.get("normal"), log_scales, fb, cfg,
                             out.get("normals_world"), touch_idx, touch_normals, None, False, intrinsics)


def adam_groups(params, grads, exp_avgs, exp_avg_sqs, lrs, step: int, beta1: float = 0.9, beta2: float = 0.999,
                eps: float = 1e-15, half_mirrors=None):
    """The argument block (fsgs_adam_groups) of the Adam step adam_step_ would launch — for launches that carry it
    (project_bin_live_count_async(adam=...)).  The tensors must stay alive until that launch has been enqueued."""
    from ._lib import AdamGroups
    n = len(params)
    assert 1 <= n <= 8
    a = AdamGroups()
    a.n_groups = n
    for k in range(n):
        a.params[k], a.grads[k] = params[k].data_ptr(), grads[k].data_ptr()
        a.exp_avg[k], a.exp_avg_sq[k] = exp_avgs[k].data_ptr(), exp_avg_sqs[k].data_ptr()
        h = half_mirrors[k] if half_mirrors is not None else None
        assert h is None or (h.dtype == torch.float16 and h.shape == params[k].shape and h.is_contiguous())
        a.half_mirror[k] = h.data_ptr() if h is not None else None
        a.numel[k] = params[k].numel()
        a.lr[k] = float(lrs[k])
    a.step, a.beta1, a.beta2, a.eps = int(step), float(beta1), float(beta2), float(eps)
    return a


def adam_step_(params, grads, exp_avgs, exp_avg_sqs, lrs, step: int, beta1: float = 0.9, beta2: float = 0.999,
               eps: float = 1e-15, half_mirrors=None) -> None:
    """In-place torch.optim.Adam update of several parameter tensors in ONE launch (row N1).  ``half_mirrors``: per
    parameter a float16 tensor of the same shape (or None) that is rewritten with the updated values in the same
    launch (BASELINE config #5: what the render kernels read)."""
    lib = load()
    n = len(params)
    dev = params[0].device
    VP = C.c_void_p * n
    numel = (C.c_int64 * n)(*[p.numel() for p in params])
    lr = (C.c_float * n)(*[float(x) for x in lrs])
    if half_mirrors is not None:
        for p_, h_ in zip(params, half_mirrors):
            assert h_ is None or (h_.dtype == torch.float16 and h_.shape == p_.shape and h_.is_contiguous())
        _run(lib.fsgs_adam_step_h16, (n, VP(*[p.data_ptr() for p in params]), VP(*[g.data_ptr() for g in grads]),
                                      VP(*[m.data_ptr() for m in exp_avgs]), VP(*[v.data_ptr() for v in exp_avg_sqs]),
                                      VP(*[(h.data_ptr() if h is not None else None) for h in half_mirrors]),
                                      numel, lr, int(step), float(beta1), float(beta2), float(eps), stream_ptr(dev)),
             "fsgs_adam_step")
        return
    _run(lib.fsgs_adam_step, (n, VP(*[p.data_ptr() for p in params]), VP(*[g.data_ptr() for g in grads]),
                              VP(*[m.data_ptr() for m in exp_avgs]), VP(*[v.data_ptr() for v in exp_avg_sqs]),
                              numel, lr, int(step), float(beta1), float(beta2), float(eps), stream_ptr(dev)),
         "fsgs_adam_step")
