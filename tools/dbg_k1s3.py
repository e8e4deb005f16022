#!/usr/bin/env python3
"""Debug aid: is the bit plane of the negative-phase sample (written by the fused K2) the same sample as its bf16 plane?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from imdbn import engine as E
from imdbn.models import RBM
F32 = np.float32
dev = torch.device("cuda:0")
eng = E.get_hip_engine()
for (V, H, B) in [(10000, 1500, 64), (2048, 512, 64)]:
    g = np.random.Generator(np.random.PCG64(3))
    W0 = (g.standard_normal((V, H), dtype=F32) * F32(0.01)).astype(F32)
    X = (g.random((B, V), dtype=F32) > 0.9).astype(F32)
    r = RBM(V, H, 0.1, 1e-4, 0.5).to(dev)
    r.W.data.copy_(torch.from_numpy(W0).to(dev)); r.W_m.zero_()
    x = torch.from_numpy(X).to(dev)
    eng.cd_factors(r, x, 1, E.PhiloxRng(seed=21), data_binary=True)
    torch.cuda.synchronize()
    Bp = 64
    V64 = (V + 63) // 64 * 64
    for which, src in (("data", "vis_tr0"), ("sample", "vis_tr1")):
        bits = eng.debug_buffer(dev, V, H, B, "vis_bits0" if which == "data" else "vis_bits1", V64 // 8 * Bp).cpu().numpy().reshape(V64 // 8, Bp)
        plane = eng.debug_buffer(dev, V, H, B, src, V * Bp * 2).cpu().numpy().view(np.uint16).reshape(V, Bp)
        want = (plane != 0)                                            # [V][Bp]
        got = np.unpackbits(bits[:, :, None], axis=2, bitorder="little")   # [V64/8][Bp][8] -> k = 8*byte + bit
        got = got.transpose(0, 2, 1).reshape(V64, Bp)[:V].astype(bool)
        bad = np.argwhere(got != want)
        print(f"{V}x{H} {which}: {len(bad)} of {want.size} bits differ; ones {want.sum()} vs {got.sum()}; bad batch rows {sorted(set(bad[:,1].tolist()))[:20]}; bad cols (first 20) {sorted(set(bad[:,0].tolist()))[:20]}")
        if len(bad):
            cols = np.array(sorted(set(bad[:, 0].tolist())))
            print("    bad columns mod 24:", sorted(set((cols % 24).tolist())), " n bad cols", len(cols))
