#!/usr/bin/env python3
"""Debug aid: k1_stream against the round-1 K1 path and the oracle (PHILOX draws), update by update."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle.rbm_oracle as O
from oracle.draws import PhiloxStream
from imdbn import engine as E
from imdbn.models import RBM
F32 = np.float32
dev = "cuda:0"
eng = E.get_hip_engine()
KEYS = ("W", "hid_bias", "vis_bias", "W_m", "hb_m", "vb_m")

def rel(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

def run(V, H, B, opts, prefetch, seed=21, updates=2, contiguous=False):
    for k, v in opts.items():
        eng.set_option(k, v)
    g = np.random.Generator(np.random.PCG64(3))
    W0 = (g.standard_normal((V, H), dtype=F32) * F32(0.01)).astype(F32)
    Xs = [(g.random((B, V), dtype=F32) > 0.9).astype(F32) for _ in range(updates)]
    r = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(dev)
    if contiguous:
        r.W.data = torch.from_numpy(W0.copy()).to(dev); r.W_m = torch.zeros_like(r.W.data)
    else:
        r.W.data.copy_(torch.from_numpy(W0).to(dev)); r.W_m.zero_()
    ts = [torch.from_numpy(x).to(dev) for x in Xs]
    st = O.RBMState.create(W0, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95)
    ps = PhiloxStream(seed)
    out = []
    with E.use_rng(E.PhiloxRng(seed=seed)):
        for i in range(updates):
            nxt = ts[i + 1] if (prefetch and i + 1 < updates) else None
            l = float(r.train_epoch(ts[i], 0, 10, CD=1, next_data=nxt))
            lo = float(O.train_epoch(st, Xs[i], 0, 1, ps))
            errs = {k: rel(getattr(r, k).detach().cpu().numpy(), getattr(st, k)) for k in KEYS}
            out.append((l, lo, errs))
    for k in opts:
        eng.set_option(k, 0)
    return out

for (V, H, B) in [(10000, 1500, 64), (2048, 512, 64), (1100, 132, 200)]:
    for name, opts, pf in [("old K1", {"no_k1s": 1}, False), ("k1s", {}, False), ("k1s+prefetch", {}, True), ("old+prefetch", {"no_k1s": 1}, True)]:
        res = run(V, H, B, opts, pf)
        for i, (l, lo, e) in enumerate(res):
            print(f"{V}x{H} B={B} {name:14s} upd {i}: loss {l:.7f} vs {lo:.7f} | " + " ".join(f"{k}={v:.1e}" for k, v in e.items()), flush=True)
