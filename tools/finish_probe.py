#!/usr/bin/env python3
"""Timing probe: one propagation repeated, in a chosen epilogue configuration (run under rocprofv3 --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import torch
import __graft_entry__ as ge
ge.build()
from imdbn import engine as E
from imdbn.models import RBM
mode = sys.argv[1]
V, H, B = 10000, 1500, 64
dev = "cuda:0"
r = RBM(V, H, 0.1, 1e-4, 0.5).to(dev)
x = (torch.rand(B, V, device=dev) > 0.9).float()
h = (torch.rand(B, H, device=dev) > 0.5).float()
eng = E.get_hip_engine()
rng = E.PhiloxRng(1)
for i in range(40):
    if mode == "up_mean":
        eng.prop_up(r, x)
    elif mode == "up_sample":
        eng.prop_up(r, x, sample=True, rng=rng)
    elif mode == "down_mean":
        eng.prop_down(r, h)
    elif mode == "down_logits":
        eng.prop_down(r, h, logits_only=True)
    elif mode == "gibbs":
        eng.gibbs_step(r, x, True, True, rng)
torch.cuda.synchronize()
