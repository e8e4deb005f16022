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
_mode = "factors"
_binary = False
_rows_ok = set()      # per-rank row counts already agreed on by all ranks (validate_rows)


def enable(group=None, force: bool = False, mode: str = "factors", binary_data: bool = False):
    """Turn on data-parallel updates (requires an initialised torch.distributed).

    ``mode``: ``"factors"`` (default) all-gathers each rank's factor block (operand planes, column sums, error
    partials: ~7 MB per 64 rows at 10000 x 1500) and every rank runs the update kernel once per rank block;
    ``"allreduce"`` all-reduces the packed fp32 statistics (60 MB at that shape).  The factor exchange needs
    <= 64 rows per rank, 16-B aligned weight rows and no softmax groups; other cases take the all-reduce path.
    ``force=True`` takes the data-parallel path even with a single rank (to exercise it on a one-GPU box).
    ``binary_data=True``: the caller guarantees that the batches handed to ``RBM.train_epoch`` are exactly 0/1 (binary
    images, e.g. ``imdbn.datasets.DeviceLoader``): their plane then crosses the wire as bits too (a batch that is not
    binary turns the update into NaN rather than being silently truncated).  The negative visible sample always does."""
    global _group, _enabled, _force, _mode, _binary
    _binary = bool(binary_data)
    import torch.distributed as dist
    if not dist.is_initialized():
        raise RuntimeError("imdbn.engine.dp.enable(): torch.distributed is not initialised")
    if mode not in ("factors", "allreduce"):
        raise ValueError("mode must be 'factors' or 'allreduce'")
    _group = group
    _enabled = True
    _force = bool(force)
    _mode = mode


def disable():
    global _group, _enabled, _force, _mode, _binary
    _group, _enabled, _force, _mode, _binary = None, False, False, "factors", False
    _rows_ok.clear()
    from . import get_rng, PhiloxRng           # the shard offset of the draw source is data-parallel state: undo it
    rng = get_rng()
    if isinstance(rng, PhiloxRng):
        rng.row0 = 0


def validate_rows(B: int, device) -> None:
    """Every rank must hold the SAME number of rows of a global batch: the Philox row offset is rank * B, the update is
    normalised by B * world and the exchanges use equal block sizes.  The first time a row count is seen, the ranks agree on it
    with one small all-reduce (all ranks meet a new count at the same step of a lock-step loader); a mismatch -- e.g. a ragged
    last batch split unevenly -- raises on every rank instead of hanging in the exchange or silently mis-normalising."""
    if B in _rows_ok or world_size() == 1:
        return
    import torch.distributed as dist
    t = torch.tensor([float(B), -float(B)], device=device if dist.get_backend(_group) == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_group)
    hi, lo = int(t[0].item()), int(-t[1].item())
    if hi != lo:
        raise RuntimeError(f"imdbn data parallelism: ranks hold between {lo} and {hi} rows of this global batch; every rank must hold the "
                           f"same number (drop the ragged last batch or pad it: imdbn.datasets.DeviceLoader(drop_last=True))")
    _rows_ok.add(B)


def binary_data() -> bool:
    return _binary


def mode() -> str:
    return _mode


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


def all_gather_blocks(out: torch.Tensor, block: torch.Tensor) -> torch.Tensor:
    """out[r] = rank r's block (out: [world, n] uint8, contiguous)."""
    import torch.distributed as dist
    dist.all_gather_into_tensor(out.view(-1), block.contiguous(), group=_group)
    return out
