"""Random-draw sources for the oracle and the parity tests.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
path (``multimodal-idbn_amd/``); only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may use it.

The reference (``/root/reference/imdbn/models/rbm.py``) draws its randomness with
``torch.rand_like`` (:125,:203,:208,:271,:333,:392,:395,:462), ``torch.randn_like``
(:346,:352) and ``torch.distributions.Categorical(...).sample()`` (:131, i.e.
``torch.multinomial``).  Parity is only well defined when both sides consume the
*same* draws, so every draw goes through one of two sources:

* :class:`DrawStream` -- a portable, seed-reproducible stream (numpy PCG64) that the
  fixture generator substitutes for the torch generators while the *unmodified*
  reference runs, and that tests re-create from the seed stored in the fixture.
  Categorical indices are taken from the fixture (``cat`` list), never re-derived.
* :class:`PhiloxStream` -- a numpy restatement of the engine's device-side
  Philox-4x32-10 counter RNG (see ``multimodal-idbn_amd/csrc/philox.hpp``), so a
  PHILOX-mode GPU run can be checked against the oracle draw for draw.
"""
from __future__ import annotations

import numpy as np


class DrawStream:
    """Sequential float32 draws from numpy PCG64; order == reference call order."""

    def __init__(self, seed: int, cat=None):
        self.seed = int(seed)
        self._g = np.random.Generator(np.random.PCG64(self.seed))
        self._cat = list(cat) if cat is not None else None
        self._cat_pos = 0
        self.log = []          # [("u"|"n"|"c", shape)] in consumption order
        self.cat_record = []   # filled by the fixture generator

    # -- float draws ---------------------------------------------------------
    def uniform(self, shape) -> np.ndarray:
        shape = tuple(int(s) for s in shape)
        self.log.append(("u", shape))
        return self._g.random(shape, dtype=np.float32)

    def normal(self, shape) -> np.ndarray:
        shape = tuple(int(s) for s in shape)
        self.log.append(("n", shape))
        return self._g.standard_normal(shape, dtype=np.float32)

    # -- categorical ---------------------------------------------------------
    def categorical(self, probs: np.ndarray) -> np.ndarray:
        """Replay the recorded index vector for the next Categorical draw."""
        if self._cat is None:
            raise RuntimeError("DrawStream has no recorded categorical draws")
        idx = np.asarray(self._cat[self._cat_pos], dtype=np.int64)
        self._cat_pos += 1
        self.log.append(("c", tuple(idx.shape)))
        if idx.shape[0] != probs.shape[0]:
            raise RuntimeError("categorical replay shape mismatch")
        return idx

    def exhausted_cat(self) -> bool:
        return self._cat is None or self._cat_pos == len(self._cat)


# ---------------------------------------------------------------------------
# Philox-4x32-10 (Salmon et al., SC'11) -- counter-based, identical to the device
# implementation.  counter = (c0, c1, c2, c3), key = (k0, k1).
# ---------------------------------------------------------------------------
_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox-4x32-10.  All inputs uint32 arrays (broadcastable)."""
    c0 = np.asarray(c0, dtype=np.uint32)
    c1 = np.asarray(c1, dtype=np.uint32)
    c2 = np.asarray(c2, dtype=np.uint32)
    c3 = np.asarray(c3, dtype=np.uint32)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    shape = np.broadcast(c0, c1, c2, c3).shape
    c0 = np.broadcast_to(c0, shape).copy()
    c1 = np.broadcast_to(c1, shape).copy()
    c2 = np.broadcast_to(c2, shape).copy()
    c3 = np.broadcast_to(c3, shape).copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK).astype(np.uint32)
            n0 = hi1 ^ c1 ^ k0
            n1 = lo1
            n2 = hi0 ^ c3 ^ k1
            n3 = lo0
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _u32_to_uniform(x: np.ndarray) -> np.ndarray:
    """[0,1) with 24 random bits, exactly as the device: (x >> 8) * 2^-24."""
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


class PhiloxStream:
    """Numpy twin of the engine's PHILOX mode.

    A logical draw tensor ``[B, N]`` number ``s`` (``s`` = stream offset, advanced by
    one per draw) has element ``(b, n)`` generated from

        key     = (seed_lo, seed_hi)
        normal : counter = (n, (row0 + b) >> 1, s_lo, s_hi) -> ``x[0..3]``; (a, c) = (x[0], x[1]) for an even global
                 row, (x[2], x[3]) for an odd one;  sqrt(-2 ln u1) * cos(2 pi u2),  u1 = ((a>>8)+1) * 2^-24, u2 = (c>>8) * 2^-24
        uniform: counter = (n, (row0 + b) >> 2, s_lo, s_hi) -> ``x[0..3]``;  (x[(row0 + b) & 3] >> 8) * 2^-24
                 (the four rows of a global 4-row group share one Philox block: csrc/common.hpp draw_uniform_rows)

    i.e. keyed on the *global* row index, so the value does not depend on tiling or on how
    the batch is sharded over ranks.
    Categorical draws use inverse-CDF on the uniform of element ``(b, group_index)``.
    """

    def __init__(self, seed: int, offset: int = 0, row0: int = 0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.offset = int(offset)
        self.row0 = int(row0)
        self.log = []

    def _uniform_block(self, shape):
        B, N = int(shape[0]), int(shape[1])
        s = self.offset
        self.offset += 1
        g = np.arange(B, dtype=np.uint64) + np.uint64(self.row0)
        cols = np.arange(N, dtype=np.uint32)[None, :]
        x = philox4x32_10(cols, (g >> np.uint64(2)).astype(np.uint32)[:, None], np.uint32(s & 0xFFFFFFFF),
                          np.uint32((s >> 32) & 0xFFFFFFFF), self.seed & 0xFFFFFFFF, (self.seed >> 32) & 0xFFFFFFFF)
        comp = (g & np.uint64(3)).astype(np.int64)[:, None]
        pick = np.where(comp == 0, x[0], np.where(comp == 1, x[1], np.where(comp == 2, x[2], x[3])))
        return _u32_to_uniform(pick.astype(np.uint32))

    def uniform(self, shape) -> np.ndarray:
        self.log.append(("u", tuple(shape)))
        return self._uniform_block(shape)

    def normal(self, shape) -> np.ndarray:
        self.log.append(("n", tuple(shape)))
        B, N = int(shape[0]), int(shape[1])
        s = self.offset
        self.offset += 1
        g = np.arange(B, dtype=np.uint64) + np.uint64(self.row0)
        cols = np.arange(N, dtype=np.uint32)[None, :]
        x = philox4x32_10(cols, (g >> np.uint64(1)).astype(np.uint32)[:, None], np.uint32(s & 0xFFFFFFFF),
                          np.uint32((s >> 32) & 0xFFFFFFFF), self.seed & 0xFFFFFFFF, (self.seed >> 32) & 0xFFFFFFFF)
        odd = (g & np.uint64(1)).astype(bool)[:, None]
        x0 = np.where(odd, x[2], x[0]).astype(np.uint32)
        x1 = np.where(odd, x[3], x[1]).astype(np.uint32)
        u1 = ((x0 >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(2.0 ** -24)
        u2 = _u32_to_uniform(x1)
        r = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
        return (r * np.cos(np.float32(2.0 * np.pi) * u2)).astype(np.float32)

    def categorical(self, probs: np.ndarray) -> np.ndarray:
        """Inverse-CDF on one uniform per row (device PHILOX-mode categorical).

        ``probs`` is the clamped, *unnormalised* group slice (reference: ``rbm.py:130``);
        idx = first j with cumsum(probs)[j] > u * sum(probs); float32 left-to-right sums.
        """
        B, g = probs.shape
        self.log.append(("c", (B,)))
        u = self.uniform_silent((B, 1))[:, 0]
        idx = np.zeros(B, dtype=np.int64)
        for b in range(B):
            tot = np.float32(0.0)
            for j in range(g):
                tot = np.float32(tot + probs[b, j])
            thr = np.float32(u[b] * tot)
            acc = np.float32(0.0)
            pick = g - 1
            for j in range(g):
                acc = np.float32(acc + probs[b, j])
                if acc > thr:
                    pick = j
                    break
            idx[b] = pick
        return idx

    def uniform_silent(self, shape):
        return self._uniform_block(shape)

    def exhausted_cat(self) -> bool:
        return True
