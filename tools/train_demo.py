#!/usr/bin/env python3
"""End-to-end run of the reference's training recipe on synthetic data at the paper's sizes (BASELINE configs 2, 3, 5):
iDBN [10000, 1500, 500] layer-wise pre-training, then iMDBN.train_joint (500 + 32 labels <-> 256, warm-up and main
phase, cross-modal reconstruction on every batch), save / load round trip.  Prints wall-clock per phase."""
import os, sys, time, tempfile
os.environ.setdefault("OMP_NUM_THREADS", "4")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import torch
from torch.utils.data import DataLoader, TensorDataset
import __graft_entry__ as ge
ge.build()
from imdbn import engine as E
from imdbn.models import iMDBN

dev = torch.device("cuda:0")
E.manual_seed(0)
N, K, B = 64 * 40, 32, 64
g = torch.Generator().manual_seed(1)
yi = torch.randint(0, K, (N,), generator=g)
proto = (torch.rand(K, 10000, generator=g) > 0.9).float()
X = (proto[yi] - (torch.rand(N, 10000, generator=g) > 0.97).float()).abs()
Y = torch.eye(K)[yi]
dl = DataLoader(TensorDataset(X.to(dev), Y.to(dev)), batch_size=B, shuffle=False)
params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95, "LEARNING_RATE_DYNAMIC": True,
          "CD": 1, "JOINT_LEARNING_RATE": 0.04, "JOINT_CD": 1, "JOINT_AUX_COND_STEPS": 30, "CROSS_GIBBS_STEPS": 50}
m = iMDBN([10000, 1500, 500], 256, params=params, dataloader=dl, val_loader=dl, device=dev, num_labels=K)

def timed(name, fn, n_batches):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:46s} {dt:7.2f} s   ({1e3 * dt / n_batches:7.3f} ms per batch)", flush=True)

E1, E2 = 3, 10
timed(f"image_idbn.train({E1})  [10000-1500-500, {N // B} batches/epoch]", lambda: m.image_idbn.train(E1), E1 * (N // B))
timed(f"train_joint({E2})  [8 warm-up + 2 main epochs]", lambda: m.train_joint(E2), E2 * (N // B))
h = m.joint_history[-1]
print("last epoch metrics:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in h.items() if k != "cd_losses"})
path = os.path.join(tempfile.mkdtemp(), "imdbn_demo.pkl")
m.save_model(path)
pl = iMDBN.load_model(path, device=dev)
print("saved / loaded:", sorted(pl.keys()) if isinstance(pl, dict) else type(pl).__name__, f"{os.path.getsize(path) / 1e6:.0f} MB")
