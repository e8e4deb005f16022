"""Helpers shared by the parity tests: fixture loading and draw-stream re-creation."""
from __future__ import annotations

import json
import math
import os

import numpy as np

from oracle.draws import DrawStream

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture:
    def __init__(self, name: str):
        z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        self.a = {k: z[k] for k in z.files}
        self.meta = json.loads(str(self.a.pop("meta")))
        self.cat = None
        if "cat_lens" in self.a:
            lens = self.a.pop("cat_lens")
            flat = self.a.pop("cat_flat")
            off = np.concatenate([[0], np.cumsum(lens)])
            self.cat = [flat[off[i]:off[i + 1]].astype(np.int64) for i in range(len(lens))]

    def __getitem__(self, k):
        return self.a[k]

    def stream(self) -> DrawStream:
        return DrawStream(self.meta["seed"], cat=self.cat)


def init_W(stream, V, H):
    return (stream.normal((V, H)) / np.float32(math.sqrt(max(1, V)))).astype(np.float32)


def rel_fro(a, b) -> float:
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    d = np.linalg.norm(a - b)
    n = np.linalg.norm(b)
    return float(d / n) if n > 0 else float(d)


def assert_close(got, want, rel=1e-5, what="", atol=0.0):
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    r = rel_fro(got, want)
    if r > rel and not np.allclose(got, want, rtol=0, atol=atol):
        diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
        worst = np.unravel_index(int(diff.argmax()), diff.shape) if diff.ndim else ()
        raise AssertionError(
            f"{what}: rel-Frobenius {r:.3e} > {rel:.1e}; max|d|={diff.max():.3e} at {worst} "
            f"(got {got[worst] if diff.ndim else got}, want {want[worst] if diff.ndim else want}); "
            f"{int((diff > 10 * max(atol, 1e-6)).sum())} elements differ by >1e-5 -- a handful of rows/cols "
            f"differing at ~lr/B scale means a Bernoulli unit flipped (SURVEY 7.3-a)")


def gpu_cd_samples(eng, dev, V, H, B):
    """The Bernoulli samples of the LAST CD-1 update the engine ran on the (V, H, B) workspace, read from its operand buffers
    (test aid imdbn_debug_ws_offset): (h0 [B, H], v' [B, V]) as bool arrays -- what oracle.rbm_oracle.set_tie_break takes."""
    import torch
    torch.cuda.synchronize()
    Bp, H64 = (B + 63) // 64 * 64, (H + 63) // 64 * 64
    hb = eng.debug_buffer(dev, V, H, B, "hid_bits", H64 // 8 * Bp).cpu().numpy().reshape(H64 // 8, Bp)
    h = np.unpackbits(hb[:, :, None], axis=2, bitorder="little").transpose(0, 2, 1).reshape(H64, Bp)[:H, :B].T.astype(bool)
    plane = eng.debug_buffer(dev, V, H, B, "vis_tr1", V * Bp * 2).cpu().numpy().view(np.uint16).reshape(V, Bp)
    return h, (plane[:, :B] != 0).T
