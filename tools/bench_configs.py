#!/usr/bin/env python3
"""Timings of the other BASELINE.json configs through the product classes (1 GPU, synthetic data).
Reference CPU figures (survey container, 8 cores) are printed beside them."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import torch
import __graft_entry__ as ge
ge.build()
from torch.utils.data import DataLoader, TensorDataset
from imdbn import engine as E
from imdbn.models import RBM, iDBN, iMDBN

dev = torch.device("cuda:0")
E.manual_seed(3)
PARAMS = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95,
          "LEARNING_RATE_DYNAMIC": True, "CD": 1, "JOINT_LEARNING_RATE": 0.04, "JOINT_CD": 1,
          "JOINT_AUX_COND_STEPS": 30, "CROSS_GIBBS_STEPS": 50}
out = {}

def timeit(fn, n, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

# C2: iDBN [10000,1500,500] interleaved batch step (2 CD-1 updates + 2 forwards), batch 64
X = (torch.rand(64 * 8, 10000) > 0.9).float()
dl = DataLoader(TensorDataset(X, torch.zeros(len(X), 1)), batch_size=64)
d = iDBN([10000, 1500, 500], dict(PARAMS), dl, dl, dev)
xb = [b[0].to(dev) for b in dl]
def c2():
    for v in xb:
        for r in d.layers:
            r.train_epoch(v, 0, 1, CD=1); v = r.forward(v)
t = timeit(c2, 5, 1) / len(xb)
out["C2_stack_ms_per_batch"] = 1e3 * t; out["C2_ref_cpu_ms_per_batch"] = 213.0

# C3: joint RBM 532<->256 with 32 softmax labels, batch 64
jr = RBM(532, 256, 0.04, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, softmax_groups=[(500, 532)]).to(dev)
z = torch.rand(64, 500, device=dev); y = torch.eye(32, device=dev)[torch.randint(0, 32, (64,), device=dev)]
vp = torch.cat([z, y], 1); vk = torch.zeros(64, 532, device=dev); km = torch.zeros(64, 532, device=dev)
vk[:, 500:] = y; km[:, 500:] = 1
def c3_main():
    jr.train_epoch(vp, 9, 20, CD=1)
    jr.train_epoch_clamped(vk, km, 9, 20, CD=1, cond_init_steps=30, sample_h=False, sample_v=False,
                           reclamp_negative=False, aux_lr_mult=0.3, use_noisy_init=True)
def c3_warm():
    for _ in range(2):
        jr.train_epoch_clamped(vk, km, 0, 20, CD=1, cond_init_steps=30, sample_h=False, sample_v=False,
                               aux_lr_mult=0.3, use_noisy_init=True)
out["C3_main_step_ms"] = 1e3 * timeit(c3_main, 20); out["C3_ref_cpu_ms"] = 30.7
out["C3_warmup_step_ms"] = 1e3 * timeit(c3_warm, 20)

# C5: _cross_reconstruct, batch 256, 50 steps, decode to 10000 px
m = iMDBN([10000, 1500, 500], 256, params=dict(PARAMS), dataloader=dl, val_loader=dl, device=dev, num_labels=32)
m.z_class_mean = torch.rand(32, 500, device=dev)
z5 = torch.rand(256, 500, device=dev); y5 = torch.eye(32, device=dev)[torch.randint(0, 32, (256,), device=dev)]
out["C5_cross_reconstruct_ms"] = 1e3 * timeit(lambda: m._cross_reconstruct(z5, y5, steps=50), 10); out["C5_ref_cpu_ms"] = 263.0
m.live_best_of_k, m.best_of_k = True, 16      # K=16 with live free-energy selection
out["C5_cross_reconstruct_live_k16_ms"] = 1e3 * timeit(lambda: m._cross_reconstruct(z5, y5, steps=50), 10)
print(json.dumps(out))
