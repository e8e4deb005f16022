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
if mode == "joint":
    jr = RBM(532, 256, 0.04, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, softmax_groups=[(500, 532)]).to("cuda:0")
    z = torch.rand(64, 500, device="cuda:0"); y = torch.eye(32, device="cuda:0")[torch.randint(0, 32, (64,), device="cuda:0")]
    vp = torch.cat([z, y], 1); vk = torch.zeros(64, 532, device="cuda:0"); km = torch.zeros(64, 532, device="cuda:0")
    vk[:, 500:] = y; km[:, 500:] = 1
    E.manual_seed(1)
    for i in range(20):
        jr.train_epoch(vp, 9, 20, CD=1)
        jr.train_epoch_clamped(vk, km, 9, 20, CD=1, cond_init_steps=30, sample_h=False, sample_v=False,
                               reclamp_negative=False, aux_lr_mult=0.3, use_noisy_init=True)
    torch.cuda.synchronize()
    sys.exit(0)
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
