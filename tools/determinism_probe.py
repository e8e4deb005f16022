#!/usr/bin/env python3
"""Run-to-run determinism under different stale workspace contents: the same 6-step CD sequence must give the same bits whatever
the scratch held before (zeros, NaN, random finite garbage, the previous run's contents)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd"), os.path.join(ROOT, "tests")]
import __graft_entry__ as ge
ge.build(compile_ok=False)
import numpy as np, torch
from imdbn import engine as E
from imdbn.models import RBM
import parity_cases as P
DEV = "cuda:0"
eng = E.get_hip_engine()
shapes = [(2048, 512, 40), (10000, 1500, 64), (784, 256, 32), (1500, 500, 64), (4099, 130, 33)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for V, H, B in shapes:
    g = np.random.default_rng(V)
    W0 = (g.standard_normal((V, H)) / np.sqrt(V)).astype(np.float32)
    Xs = [P.T((g.random((B, V), dtype=np.float32) > 0.7).astype(np.float32), DEV) for _ in range(6)]
    ref = None
    for rep in range(reps):
        r = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95)
        P.set_params(r, DEV, W0, np.zeros(H, np.float32), np.zeros(V, np.float32))
        ws = eng._workspace(torch.device(DEV), V, H, B)
        kind = rep % 4
        if kind == 0: ws.zero_()
        elif kind == 1: ws.view(torch.float32).fill_(float("nan"))
        elif kind == 2: ws.view(torch.int32).random_(-(1 << 30), 1 << 30)
        with E.use_rng(E.PhiloxRng(seed=12)):
            ls = [float(r.train_epoch(x, 2, 10, CD=1, next_data=Xs[i + 1] if (i + 1 < 6 and rep % 2) else None)) for i, x in enumerate(Xs)]
        cur = (ls, r.W.data.clone(), r.W_m.clone(), r.hid_bias.data.clone(), r.vis_bias.data.clone())
        if ref is None:
            ref = cur
        else:
            same = ref[0] == cur[0] and all(torch.equal(a, b) for a, b in zip(ref[1:], cur[1:]))
            if not same:
                bad += 1
                d = [float((a - b).abs().max()) for a, b in zip(ref[1:], cur[1:])]
                print(f"MISMATCH {V}x{H} B={B} rep {rep} (stale kind {kind}, prefetch {rep % 2}): losses equal {ref[0] == cur[0]}, max abs diffs W/W_m/hb/vb {d}", flush=True)
    print(f"{V}x{H} B={B}: {reps} repeats done", flush=True)
print("determinism probe:", "OK" if bad == 0 else f"{bad} mismatches")
sys.exit(1 if bad else 0)
