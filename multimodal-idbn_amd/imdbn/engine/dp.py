"""Data-parallel state: one process per GPU, one all-reduce of the packed CD statistics per update.

The reference has no parallelism of any kind; this is the capability north_star
adds.  Semantics: every rank holds a full replica of the parameters, runs the
positive/negative phase on its rows, contributes un-normalised statistics
``[dW, dc, db, sum P+, sq-err]`` in ONE packed fp32 buffer, the buffer is all-reduced (RCCL over xGMI
through ``torch.distributed``'s ``nccl`` backend; ``gloo`` in the CPU tests), and every rank applies
the identical update with ``1/global_batch`` -- replicas stay bit-identical.
"""
from __future__ import annotations

import torch

_group = None
_enabled = False
_force = False


def enable(group=None, force: bool = False):
    """Turn on data-parallel updates (requires an initialised torch.distributed).

    ``force=True`` takes the stats -> all-reduce -> apply path even with a single rank (used to
    exercise the RCCL path on a one-GPU box)."""
    global _group, _enabled, _force
    import torch.distributed as dist
    if not dist.is_initialized():
        raise RuntimeError("imdbn.engine.dp.enable(): torch.distributed is not initialised")
    _group = group
    _enabled = True
    _force = bool(force)


def disable():
    global _group, _enabled, _force
    _group, _enabled, _force = None, False, False


def active() -> bool:
    if not _enabled:
        return False
    import torch.distributed as dist
    return dist.is_initialized() and (dist.get_world_size(_group) > 1 or _force)


def world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size(_group) if (_enabled and dist.is_initialized()) else 1


def rank() -> int:
    import torch.distributed as dist
    return dist.get_rank(_group) if (_enabled and dist.is_initialized()) else 0


def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_group)
    return t
