"""The collectives of the data-parallel step (the reference's hook for them is DDP's bucketed NCCL
all-reduce, /root/reference/dn_splatter/dn_pipeline.py:162-167) — ONE code path for every backend.

RCCL (backend "nccl") carries them between the GPUs of a node; gloo carries the SAME calls in the CPU tests and for
ranks that share one GPU (``FSGS_DIST_BACKEND=gloo``): this build's gloo implements ``all_reduce(AVG)``,
``reduce_scatter_tensor(AVG)``, the in-place ``all_gather_into_tensor`` and their asynchronous forms
(tests/test_host_logic.py: test_collectives_single_code_path_gloo_world2), so there is no per-backend branch any more —
what the world-2 / world-8 gloo tests run is, call for call, what an 8-GPU RCCL job runs (round 4 took the other side of
every ``if backend == "nccl"`` in those tests)."""
from __future__ import annotations

import os
from typing import Callable, Optional

import torch
import torch.distributed as dist
from torch import Tensor

# FSGS_FORCE_COLLECTIVES=1: issue the collectives in a one-rank group too (they are identities there) — lets a one-GPU
# box run the very RCCL calls of the multi-GPU step (tests/test_gpu_configs.py)
FORCE = os.environ.get("FSGS_FORCE_COLLECTIVES", "0") == "1"


def world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def exchange(group=None) -> bool:
    """Whether a collective has to be issued at all."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return world(group) > 1 or FORCE


def all_reduce_mean_(t: Tensor, group=None, async_op: bool = False) -> Optional[Callable[[], None]]:
    """Mean over the ranks, in place (averaged inside the collective: no scaling pass over 236 B per Gaussian).
    Returns None (done, in stream order) or a callable that completes it."""
    if not exchange(group) or t.numel() == 0:
        return None
    work = dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group, async_op=async_op)
    return work.wait if async_op else None


def reduce_scatter_mean(out: Tensor, full: Tensor, group=None) -> None:
    """out = this rank's 1/W shard of the ranks' mean of ``full`` (``full.numel() == W * out.numel()``)."""
    dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.AVG, group=group)


def all_gather_shards_(flat: Tensor, lo: int, hi: int, group=None) -> None:
    """Every rank's own slice [lo, hi) of ``flat`` lands in all ranks' ``flat`` (in place: the input IS the output's
    own slice; ``flat.numel() == W * (hi - lo)``, slices in rank order)."""
    dist.all_gather_into_tensor(flat, flat[lo:hi], group=group)


def all_gather_blocks_async(gathered: Tensor, own: Tensor, group=None):
    """gathered [W, ...] <- every rank's ``own`` block, in rank order; returns the work handle."""
    return dist.all_gather_into_tensor(gathered.view(-1), own.reshape(-1), group=group, async_op=True)


def all_reduce_(t: Tensor, op: str, group=None) -> None:
    """SUM / MAX over the ranks, in place (densification statistics, once per refinement)."""
    dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX}[op], group=group)
