#!/usr/bin/env python3
"""Debug aid: the hidden sample (bit plane) of the positive phase against p > u with the numpy Philox twin."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from oracle.draws import PhiloxStream
from imdbn import engine as E
from imdbn.models import RBM
F32 = np.float32
dev = torch.device("cuda:0")
eng = E.get_hip_engine()
V, H, B = 10000, 1500, 64
g = np.random.Generator(np.random.PCG64(3))
W0 = (g.standard_normal((V, H), dtype=F32) * F32(0.01)).astype(F32)
X = (g.random((B, V), dtype=F32) > 0.9).astype(F32)
def bf(x): return (x.astype(np.uint32) << 16).view(np.float32)
for name, opts in (("old", {"no_k1s": 1}), ("k1s", {})):
    for k, v in opts.items(): eng.set_option(k, v)
    r = RBM(V, H, 0.1, 1e-4, 0.5).to(dev)
    r.W.data.copy_(torch.from_numpy(W0).to(dev)); r.W_m.zero_()
    eng.cd_factors(r, torch.from_numpy(X).to(dev), 1, E.PhiloxRng(seed=21), data_binary=True)
    torch.cuda.synchronize()
    Bp, H64 = 64, (H + 63) // 64 * 64
    hb = eng.debug_buffer(dev, V, H, B, "hid_bits", H64 // 8 * Bp).cpu().numpy().reshape(H64 // 8, Bp)
    got = np.unpackbits(hb[:, :, None], axis=2, bitorder="little").transpose(0, 2, 1).reshape(H64, Bp)[:H].T.astype(bool)   # [B][H]
    ht = eng.debug_buffer(dev, V, H, B, "hid_tr0", 3 * H * Bp * 2).cpu().numpy().view(np.uint16).reshape(3, H, Bp)
    p = ((bf(ht[0]) + bf(ht[1])) + bf(ht[2])).T
    u = PhiloxStream(21).uniform((B, H))
    want = p > u
    bad = np.argwhere(got != want)
    print(f"{name}: {len(bad)} hidden samples differ from p > u; rows {sorted(set(bad[:,0].tolist()))[:20]}; cols(first 12) {sorted(set(bad[:,1].tolist()))[:12]}; ones got {got.sum()} want {want.sum()}")
    if len(bad):
        bb, nn = bad[0]
        print("   first bad:", bb, nn, "p", p[bb, nn], "u", u[bb, nn], "got", got[bb, nn])
        # is `got` the sample of ANOTHER row's uniform?
        for shift in range(-8, 9):
            ok = (p[0:8] > np.roll(u, shift, axis=0)[0:8])
            print("   row shift", shift, "mismatches in rows 0-7:", int((ok != got[0:8]).sum()))
    for k in opts: eng.set_option(k, 0)
