#!/usr/bin/env python3
"""Debug aid: hidden column-sum partials left by cd_factors, k1_stream vs the round-1 K1, same draws."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from imdbn import engine as E
from imdbn.models import RBM
F32 = np.float32
dev = "cuda:0"
eng = E.get_hip_engine()
V, H, B = 10000, 1500, 64
def up(x, m=256): return (x + m - 1) // m * m
Bp, P = 64, 8
Vpad = (V + 15) // 16 * 16
off_flags = 0
off_hid0 = up(P * ((Vpad + 63) // 64) * 4)
off_hid1 = off_hid0 + up(3 * H * Bp * 2)
off_cshp = off_hid1 + up(3 * H * Bp * 2)
off_cshn = off_cshp + up(P * H * 4)
g = np.random.Generator(np.random.PCG64(3))
W0 = (g.standard_normal((V, H), dtype=F32) * F32(0.01)).astype(F32)
X = (g.random((B, V), dtype=F32) > 0.9).astype(F32)
def run(opts, mode):
    for k, v in opts.items(): eng.set_option(k, v)
    r = RBM(V, H, 0.1, 1e-4, 0.5).to(dev)
    r.W.data.copy_(torch.from_numpy(W0).to(dev)); r.W_m.zero_()
    x = torch.from_numpy(X).to(dev)
    rng = E.PhiloxRng(seed=21) if mode == "philox" else None
    blk = eng.cd_factors(r, x, 1, E.PhiloxRng(seed=21), data_binary=True)
    torch.cuda.synchronize()
    b = blk.cpu().numpy()
    csp = b[off_cshp:off_cshp + P * H * 4].view(np.float32).reshape(P, H).copy()
    csn = b[off_cshn:off_cshn + P * H * 4].view(np.float32).reshape(P, H).copy()
    hp = b[off_hid0:off_hid0 + 3 * H * Bp * 2].view(np.uint16).reshape(3, H, Bp).copy()
    for k in opts: eng.set_option(k, 0)
    return csp, csn, hp
for trial in range(3):
    a = run({"no_k1s": 1}, "philox")
    b = run({}, "philox")
    for name, x, y in (("cs_hpos", a[0], b[0]), ("cs_hneg", a[1], b[1])):
        d = np.abs(x - y)
        bad = np.argwhere(d > 1e-4)
        print(f"trial {trial} {name}: max|d| {d.max():.3e}, {len(bad)} of {d.size} entries differ > 1e-4; bad columns (first 40): {sorted(set(bad[:,1].tolist()))[:40]}; bad row groups: {sorted(set(bad[:,0].tolist()))}")
    hd = (a[2] != b[2])
    cols = np.argwhere(hd.any(axis=(0, 2))).ravel()
    print(f"   hid_tr[0] planes: {hd.sum()} of {hd.size} bf16 differ; columns {cols[:40].tolist()} ... n={len(cols)}")
