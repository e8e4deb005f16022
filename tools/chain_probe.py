#!/usr/bin/env python3
"""Time the joint-RBM chains (C3 shapes): row-parallel chain kernel vs one launch per half step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import torch
import __graft_entry__ as ge
ge.build()
from imdbn import engine as E
from imdbn.models import RBM
dev = torch.device("cuda:0")
eng = E.get_hip_engine()
E.manual_seed(3)
GROUPS = None if os.environ.get("NO_GROUP") else [(500, 532)]
for B in (64, 256):
    jr = RBM(532, 256, 0.04, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, softmax_groups=GROUPS).to(dev)
    y = torch.eye(32, device=dev)[torch.randint(0, 32, (B,), device=dev)]
    vk = torch.zeros(B, 532, device=dev); km = torch.zeros(B, 532, device=dev)
    vk[:, 500:] = y; km[:, 500:] = 1
    for nk in (1, 0):
        eng.set_option("no_chain_kernel", nk)
        for name, fn in (("noisy_meanfield_annealed(30)", lambda: jr.noisy_meanfield_annealed(vk, km, n_steps=30)),
                         ("conditional_gibbs(50, sample_h, sample_v)", lambda: jr.conditional_gibbs(vk, km, n_steps=50, sample_h=True, sample_v=True))):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): fn()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
            print(f"B={B:3d} {'per-launch' if nk else 'chain kernel':12s} {name:44s} {1e3*dt:7.3f} ms")
eng.set_option("no_chain_kernel", 0)
