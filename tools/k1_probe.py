#!/usr/bin/env python3
"""Per-block timelines of k1_stream alone (RBM.forward = preparation + K1) for the operand kinds: asserted 0/1 (bit planes),
unknown 0/1 (per-item choice, bit-plane loop), real values announced / not announced (loop over bf16 terms)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import torch
import __graft_entry__ as ge
ge.build()
from imdbn import engine as E
from imdbn.engine import native
from imdbn.models import RBM

shapes = [(10000, 1500, 64), (1500, 500, 64)] if len(sys.argv) < 2 else [tuple(map(int, sys.argv[1].split("x")))]
dev = torch.device("cuda")
eng = E.get_hip_engine()
for V, H, B in shapes:
    rbm = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(dev)
    xb = (torch.rand(B, V) > 0.9).float().to(dev)
    xr = torch.rand(B, V).to(dev)
    cases = []
    t = xb.clone(); t._imdbn_binary = True; cases.append(("0/1 asserted", t))
    cases.append(("0/1 unknown", xb.clone()))
    t = xr.clone(); t._imdbn_binary = False; cases.append(("real announced", t))
    cases.append(("real unknown", xr.clone()))
    for name, x in cases:
        eng.set_option("dbg", 64)
        for i in range(10):
            rbm.forward(x)
        torch.cuda.synchronize()
        buf = (C.c_longlong * (4096 * 8))()
        native.check(native.lib().imdbn_debug_stamps(buf, 4096 * 8), "imdbn_debug_stamps")
        eng.set_option("dbg", 0)
        a = np.frombuffer(buf, dtype=np.int64).reshape(4096, 8).copy()
        nb = int((a[:, 0] > 0).sum())
        t0 = a[:nb, 0].min()
        last = a[:nb][a[:nb, 6] > 0]
        b = (last[:, :7] - t0).astype(np.float64) / 100.0
        allb = (a[:nb, :5] - t0).astype(np.float64) / 100.0
        # wall time of the pair of launches (prep + K1) over 200 calls
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for i in range(200):
            rbm.forward(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t1) / 200
        print(f"== {V}x{H} B={B} {name:16s}: {nb} blocks, forward() {1e6 * dt:6.1f} us/call; all blocks p50 [start, landed, loop, reduced, published] "
              + " ".join(f"{np.percentile(allb[:, j], 50):6.2f}" for j in range(5))
              + " | last arrivers max [.., combined, epilogue] " + " ".join(f"{b[:, j].max():6.2f}" for j in range(7)))
