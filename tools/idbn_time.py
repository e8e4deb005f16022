#!/usr/bin/env python3
"""iDBN.train wall clock per batch on the paper-size stack through a stock DataLoader (in-memory TensorDataset on the GPU),
first epoch (one-time costs) and steady state."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import __graft_entry__ as ge
ge.build(compile_ok=False)
import torch
from torch.utils.data import DataLoader, TensorDataset
from imdbn import engine as E
from imdbn.models import iDBN
dev = torch.device("cuda:0")
E.manual_seed(0)
os.chdir(tempfile.mkdtemp())
N, B = 64 * 40, 64
X = (torch.rand(N, 10000) > 0.9).float()
for where in ("cuda", "cpu"):
    dl = DataLoader(TensorDataset(X.to(where), torch.zeros(N, 1).to(where)), batch_size=B, shuffle=False)
    params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95, "LEARNING_RATE_DYNAMIC": True, "CD": 1}
    d = iDBN([10000, 1500, 500], params, dl, dl, dev)
    for name, ep in (("first epoch", 1), ("next 3 epochs", 3)):
        torch.cuda.synchronize(); t0 = time.perf_counter(); d.train(ep); torch.cuda.synchronize()
        print(f"dataset on {where}: {name}: {1e3 * (time.perf_counter() - t0) / (ep * N // B):.3f} ms per batch", flush=True)
