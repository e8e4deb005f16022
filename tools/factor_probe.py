#!/usr/bin/env python3
"""Cost of the data-parallel update from gathered factor blocks (imdbn_rbm_apply_factors) for R emulated ranks."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import torch
import __graft_entry__ as ge
ge.build()
from imdbn import engine as E
from imdbn.models import RBM
dev = torch.device("cuda:0"); eng = E.get_hip_engine()
V, H, B = 10000, 1500, 64
rbm = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(dev)
x = (torch.rand(B, V) > 0.9).float().to(dev)
for R in (1, 2, 4, 8):
    g = eng.gather_buffer(rbm, B, R)
    for rk in range(R):
        g[rk].copy_(eng.cd_factors(rbm, x, 1, E.PhiloxRng(seed=1, row0=rk * B)))
    for _ in range(5): eng.apply_factors(rbm, g, B, B * R, 0.01, 0.5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): eng.apply_factors(rbm, g, B, B * R, 0.01, 0.5)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    t1 = time.perf_counter()
    for _ in range(50): eng.cd_factors(rbm, x, 1, E.PhiloxRng(seed=1))
    torch.cuda.synchronize(); dc = (time.perf_counter() - t1) / 50
    print(f"R={R}: apply_factors {1e6*dt:7.1f} us   (cd_factors {1e6*dc:6.1f} us; block {g.size(1)/1e6:.2f} MB per rank)")
