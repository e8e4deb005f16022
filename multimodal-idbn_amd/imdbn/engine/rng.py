"""Host-side description of where the engine's random draws come from.

The reference draws with ``torch.rand_like`` / ``torch.randn_like`` / ``Categorical.sample``
(rbm.py:125,131,203,208,271,333,346,352,392,395,462).  The engine never calls torch's generators;
it consumes draws in the *same order*  from one of:

``PhiloxRng``   device-side Philox-4x32-10 keyed on (seed, draw number, global row, column).
``ReplayRng``   caller-provided draws (a provider with ``uniform(shape)``, ``normal(shape)``,
                ``categorical(probs)`` returning numpy arrays) uploaded as a tape -- used by the
                parity tests to feed the draws recorded from the reference.

A *schedule* is the ordered list of draw tensors one engine call consumes:
``("u", N)`` uniform [B,N], ``("n", N)`` normal [B,N], ``("c", width)`` one categorical index per row.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

Schedule = List[Tuple[str, int]]


class PhiloxRng:
    def __init__(self, seed: int | None = None, row0: int = 0):
        self.seed = int(torch.initial_seed() if seed is None else seed) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0          # number of draw tensors consumed so far
        self.row0 = int(row0)    # global index of local row 0 (data-parallel shard offset)
        # graph capture (imdbn.engine.graph.CapturedSteps): a 1-element int64 device tensor the kernels add to `offset`
        # when they RUN; set only while a capture is being recorded
        self.device_counter = None

    def advance(self, n_draws: int):
        self.offset += int(n_draws)


class ReplayRng:
    def __init__(self, provider):
        self.provider = provider

    def build(self, schedule: Schedule, B: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
        floats, cats = [], []
        for kind, n in schedule:
            if kind == "u":
                floats.append(np.ascontiguousarray(self.provider.uniform((B, n)), dtype=np.float32).ravel())
            elif kind == "n":
                floats.append(np.ascontiguousarray(self.provider.normal((B, n)), dtype=np.float32).ravel())
            elif kind == "c":
                idx = np.asarray(self.provider.categorical(np.zeros((B, n), np.float32)), dtype=np.int32)
                if idx.shape != (B,):
                    raise ValueError("categorical replay must yield one index per row")
                cats.append(idx)
            else:
                raise ValueError(kind)
        ft = torch.from_numpy(np.concatenate(floats) if floats else np.zeros(1, np.float32)).to(device)
        ct = torch.from_numpy(np.concatenate(cats) if cats else np.zeros(1, np.int32)).to(device)
        return ft, ct

    def advance(self, n_draws: int):
        pass


# ---- schedules: must mirror the consumption order in csrc/engine.hip ---------------------------
def sched_sample_visible(V: int, groups: Sequence[Tuple[int, int]]) -> Schedule:
    return [("u", V)] + [("c", e - s) for s, e in groups]


def sched_cd(V: int, H: int, groups, cd_k: int) -> Schedule:
    s: Schedule = [("u", H)]
    for _ in range(int(cd_k)):
        s += sched_sample_visible(V, groups) + [("u", H)]
    return s


def sched_chain(V: int, H: int, groups, steps, init_uniform: bool) -> Schedule:
    s: Schedule = [("u", V)] if init_uniform else []
    for st in steps:
        if st["sigma"] > 0:
            s.append(("n", H))
        if st["sample_h"]:
            s.append(("u", H))
        if st["sigma"] > 0:
            s.append(("n", V))
        if st["vmode"] != 0:
            s += sched_sample_visible(V, groups)
    return s


def sched_clamped(V: int, H: int, groups, init_steps, cd_k: int, sample_h: bool, sample_v: bool) -> Schedule:
    s = sched_chain(V, H, groups, init_steps, True)
    for _ in range(int(cd_k)):
        if sample_h:
            s.append(("u", H))
        if sample_v:
            s += sched_sample_visible(V, groups)
    return s
