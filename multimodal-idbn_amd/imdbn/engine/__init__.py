"""Engine selection and global RNG state for the CD hot path.

Product rule: tensors on a HIP device go to :class:`HipEngine` (native kernels); anything else
raises -- there is deliberately NO CPU implementation in the product.  Tests may inject a test
double with :func:`set_engine_for_testing` to exercise the host-side logic on a CPU-only box.
"""
from __future__ import annotations

import contextlib

import torch

from . import dp  # noqa: F401
from .native import EngineError
from .rng import PhiloxRng, ReplayRng


def __getattr__(name):          # imdbn.engine.graph / CapturedSteps on demand (imports torch.cuda pieces)
    if name == "CapturedSteps":
        from .graph import CapturedSteps
        return CapturedSteps
    raise AttributeError(name)

_hip = None
_override = None
_rng = None


def set_engine_for_testing(engine):
    """Install (or with ``None`` remove) a test double implementing the HipEngine interface."""
    global _override
    _override = engine


def get_engine(t: torch.Tensor):
    global _hip
    if _override is not None:
        return _override
    if not t.is_cuda:
        raise EngineError(
            f"imdbn: tensor on {t.device}; this build runs the contrastive-divergence path only on an AMD GPU "
            f"through libimdbn_hip.so and has no CPU fallback. Move the model and data to 'cuda'.")
    if _hip is None:
        from .hip_engine import HipEngine
        _hip = HipEngine()
    return _hip


def get_hip_engine():
    """The process-wide HipEngine (loads the native library; raises if it is not built)."""
    global _hip
    if _hip is None:
        from .hip_engine import HipEngine
        _hip = HipEngine()
    return _hip


def get_rng():
    """Current draw source; defaults to Philox seeded from ``torch.initial_seed()``."""
    global _rng
    if _rng is None:
        _rng = PhiloxRng()
    return _rng


def set_rng(rng):
    global _rng
    _rng = rng


def manual_seed(seed: int, row0: int = 0):
    set_rng(PhiloxRng(seed, row0=row0))


@contextlib.contextmanager
def use_rng(rng):
    global _rng
    old = _rng
    _rng = rng
    try:
        yield rng
    finally:
        _rng = old


__all__ = ["get_engine", "get_hip_engine", "set_engine_for_testing", "get_rng", "set_rng", "manual_seed", "use_rng",
           "PhiloxRng", "ReplayRng", "EngineError", "dp"]
