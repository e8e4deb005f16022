#!/usr/bin/env python3
"""Per-block timelines of K1 / K2 (debug stamps): where does a 16 us kernel spend its time?"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import torch
import __graft_entry__ as ge
ge.build()
from imdbn import engine as E
from imdbn.engine import native
from imdbn.models import RBM

V, H, B = 10000, 1500, 64
dev = torch.device("cuda")
eng = E.get_hip_engine()
rbm = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(dev)
xs = [(torch.rand(B, V) > 0.9).float().to(dev) for _ in range(4)]
if "--tag" in sys.argv:
    for t in xs: t._imdbn_binary = True
E.set_rng(E.PhiloxRng(seed=2))
for which, bit, nslots in (("K1 k1_stream POSITIVE phase: start, bits+ring landed, loop done, reduced, published, [last arriver:] combined, epilogue done", 2048, 7),
                           ("K1 k1_stream (last launch = negative phase): start, bits+ring landed, loop done, reduced, published, [last arriver:] combined, epilogue done", 64, 7),
                           ("K2 k2_stream: start, bits staged, loop done, reduced, epilogue done, loss done", 128, 6),
                           ("K3 assoc_update_planes", 512, 6)):
    eng.set_option("dbg", bit)
    for i in range(20):
        rbm.train_epoch(xs[i % 4], 0, 1, CD=1, next_data=xs[(i + 1) % 4])
    torch.cuda.synchronize()
    buf = (C.c_longlong * (4096 * 8))()
    native.check(native.lib().imdbn_debug_stamps(buf, 4096 * 8), "imdbn_debug_stamps")
    a = np.frombuffer(buf, dtype=np.int64).reshape(4096, 8).copy()
    nb = int((a[:, 0] > 0).sum())
    if bit in (64, 2048):      # k1_stream: only the last arriver of a tile has slots 5, 6
        w7 = (a[:nb, 7] - a[:nb, 2]).astype(np.float64) / 100.0
        print(f"== k1_stream: the youngest wave leaves the K loop after wave 0 by (us): p10 {np.percentile(w7, 10):.2f} p50 {np.percentile(w7, 50):.2f} p90 {np.percentile(w7, 90):.2f} max {w7.max():.2f}")
        last = a[:nb][a[:nb, 6] > 0]
        b = last[:, :7].astype(np.float64) / 100.0
        b -= a[:nb, 0].min() / 100.0
        print(f"== k1_stream last arrivers ({len(b)} blocks): p50 " + " ".join(f"{np.percentile(b[:, j], 50):7.2f}" for j in range(7)) + " | max " + " ".join(f"{b[:, j].max():7.2f}" for j in range(7)))
        nslots = 5
    a = a[:nb, :nslots].astype(np.float64) / 100.0     # us
    t0 = a[:, 0].min()
    a -= t0
    print(f"== {which}: {nb} blocks; columns = stamp slots (us since first block start)")
    for q, nm in ((0, "min"), (50, "p50"), (90, "p90"), (100, "max")):
        print(f"   {nm:4s} " + " ".join(f"{np.percentile(a[:, j], q):7.2f}" for j in range(nslots)))
    if nslots == 8:
        print("   K2: slot7 (after per-row loop) - slot4 (epilogue start) p50: %.2f ; slot5 - slot7: %.2f" % (np.percentile(a[:, 7] - a[:, 4], 50), np.percentile(a[:, 5] - a[:, 7], 50)))
        a = a[:, :7]; nslots = 7
    d = np.diff(a, axis=1)
    print("   phase durations p50: " + " ".join(f"{np.percentile(d[:, j], 50):7.2f}" for j in range(nslots - 1)))
    print("   phase durations max: " + " ".join(f"{d[:, j].max():7.2f}" for j in range(nslots - 1)))
eng.set_option("dbg", 0)
