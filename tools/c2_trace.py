#!/usr/bin/env python3
"""The C2 stack (iDBN [10000,1500,500], batch 64) alone, for a rocprofv3 kernel trace: which launches a batch consists of."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import __graft_entry__ as ge
ge.build(compile_ok=False)
import torch
from torch.utils.data import DataLoader, TensorDataset
from imdbn import engine as E
from imdbn.models import iDBN
dev = torch.device("cuda")
E.manual_seed(3)
params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95,
          "LEARNING_RATE_DYNAMIC": True, "CD": 1}
os.chdir(tempfile.mkdtemp())
X = (torch.rand(64 * 8, 10000) > 0.9).float()
dl = DataLoader(TensorDataset(X, torch.zeros(len(X), 1)), batch_size=64)
d = iDBN([10000, 1500, 500], dict(params), dl, dl, dev)
xb = [b[0].to(dev) for b in dl]
def c2():
    last = len(d.layers) - 1
    for i, v in enumerate(xb):
        for li, r in enumerate(d.layers):
            nd = xb[(i + 1) % len(xb)] if li == 0 else None
            if li < last:
                _, v = r.train_epoch(v, 0, 1, CD=1, next_data=nd, return_forward=True)
            else:
                r.train_epoch(v, 0, 1, CD=1, next_data=nd)
for _ in range(2): c2()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for _ in range(n): c2()
torch.cuda.synchronize()
print("C2 stack: %.1f us per batch" % ((time.perf_counter() - t0) / n / len(xb) * 1e6))
