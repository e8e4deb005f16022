"""hipGraph capture of a sequence of engine calls (launch-bound inner loops: small layers, the joint RBM).

Every engine call is a plain sequence of kernel launches on the caller's stream, so a training step -- or the whole layer loop
of one batch -- can be recorded once and replayed with one graph launch instead of one Python -> C -> HIP round trip per call
(~45 us of host time per ``train_epoch``).  What changes from replay to replay is handled like this:

* **random draws**: the kernels of a captured call read a device-resident counter that a node at the end of the graph advances by
  the number of draw tensors the sequence consumes (``imdbn_rng.dev_offset`` / ``imdbn_rng_advance``), so replay k draws exactly
  what the k-th eager call would have drawn -- results are bit-identical to the eager sequence;
* **the batch**: the captured calls read STATIC input tensors; copy each new batch into them (``x_static.copy_(batch)``) before
  ``replay`` (what a batch contains is found out on the device, so nothing about it has to be declared);
* **learning rate / momentum** are host scalars of the epoch: capture once per epoch (``CapturedSteps.epoch`` is a convenience).

The next-batch prefetch (``next_data=``) names a second tensor by address and is left out of captured sequences.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import get_hip_engine, get_rng
from .rng import PhiloxRng


class CapturedSteps:
    """``fn()`` = any sequence of engine calls on CUDA tensors that draws from ``rng`` (default: the current draw source).

    The first ``__call__`` runs ``fn`` eagerly (warm-up: workspaces, one-time kernel attributes), the second records the graph and
    runs it, every later one replays it.  Returns what ``fn`` returned (tensors of the captured run: the same storage is
    overwritten by every replay -- clone what must survive)."""

    def __init__(self, fn: Callable[[], object], rng: Optional[PhiloxRng] = None):
        self.fn = fn
        self.rng = rng if rng is not None else get_rng()
        if not isinstance(self.rng, PhiloxRng):
            raise TypeError("CapturedSteps needs a PhiloxRng (device-side draws)")
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.counter: Optional[torch.Tensor] = None
        self.out = None
        self._calls = 0
        self._baked = 0          # rng.offset the captured kernels were recorded with
        self._draws = 0          # draw tensors one replay consumes
        self._shadow = 0         # host copy of the device counter

    def __call__(self):
        self._calls += 1
        if self._calls == 1:
            return self.fn()                               # eager warm-up (a real step)
        if self.graph is None:
            return self._capture_and_run()
        return self.replay()

    def _capture_and_run(self):
        eng = get_hip_engine()
        dev = torch.device("cuda", torch.cuda.current_device())
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self._baked, self._shadow = self.rng.offset, 0
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        self.rng.device_counter = self.counter
        try:
            with torch.cuda.graph(g):
                self.out = self.fn()
                self._draws = self.rng.offset - self._baked   # the calls advanced the host cursor while being recorded
                eng.rng_advance(self.counter, self._draws)
        finally:
            self.rng.device_counter = None
        self.graph = g
        self.rng.offset = self._baked                      # nothing ran yet: the first replay is the recorded step
        return self.replay()

    def replay(self):
        want = self.rng.offset - self._baked               # eager calls in between may have moved the host cursor
        if want != self._shadow:
            self.counter.fill_(want)
            self._shadow = want
        self.graph.replay()
        self._shadow += self._draws
        self.rng.offset += self._draws
        return self.out
