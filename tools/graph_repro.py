#!/usr/bin/env python3
"""Repeat the captured-vs-eager comparison with a perturbed allocator; report the first step whose weights differ."""
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
V, H, B = (int(x) for x in os.environ.get("SHAPE", "2048,512,40").split(","))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = np.random.default_rng(V)
W0 = (g.standard_normal((V, H)) / np.sqrt(V)).astype(np.float32)
Xs = [P.T((g.random((B, V), dtype=np.float32) > 0.7).astype(np.float32), DEV) for _ in range(7)]
def mk():
    r = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95)
    P.set_params(r, DEV, W0, np.zeros(H, np.float32), np.zeros(V, np.float32)); return r
ra = mk(); ref = []
with E.use_rng(E.PhiloxRng(seed=12)):
    for x in Xs:
        l = float(ra.train_epoch(x, 2, 10, CD=1)); ref.append((l, ra.W.data.clone(), ra.W_m.clone(), ra.hid_bias.data.clone()))
bad = 0
junk = []
for rep in range(reps):
    junk.append(torch.randn(int(np.random.randint(1, 64)) << 18, device=DEV))      # perturb the allocator / leave garbage around
    if len(junk) > 6: junk.pop(0)
    rb = mk()
    xs = Xs[0].clone(); xs._imdbn_binary = True
    with E.use_rng(E.PhiloxRng(seed=12)):
        step = E.CapturedSteps(lambda: rb.train_epoch(xs, 2, 10, CD=1))
        for i, x in enumerate(Xs):
            xs.copy_(x)
            l = float(rb.train_epoch(xs, 2, 10, CD=1)) if i == 4 else float(step())
            ok = l == ref[i][0] and torch.equal(rb.W.data, ref[i][1]) and torch.equal(rb.W_m, ref[i][2]) and torch.equal(rb.hid_bias.data, ref[i][3])
            if not ok:
                bad += 1
                print(f"rep {rep} step {i}: loss equal {l == ref[i][0]}; max|dW| {float((rb.W.data - ref[i][1]).abs().max()):.3e} n_diff {int((rb.W.data != ref[i][1]).sum())} "
                      f"max|dWm| {float((rb.W_m - ref[i][2]).abs().max()):.3e} max|dhb| {float((rb.hid_bias.data - ref[i][3]).abs().max()):.3e}", flush=True)
                break
    del step
print("graph repro:", "OK" if bad == 0 else f"{bad} of {reps} runs diverged")
