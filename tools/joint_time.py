#!/usr/bin/env python3
"""iMDBN.train_joint wall clock per batch (paper sizes, synthetic data): warm-up epochs and main epochs separately."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import __graft_entry__ as ge
ge.build(compile_ok=False)
import torch
from torch.utils.data import DataLoader, TensorDataset
from imdbn import engine as E
from imdbn.models import iMDBN
import imdbn.models.imdbn as IM
dev = torch.device("cuda:0")
E.manual_seed(0)
os.chdir(tempfile.mkdtemp())
N, K, B = 64 * 20, 32, 64
g = torch.Generator().manual_seed(1)
yi = torch.randint(0, K, (N,), generator=g)
X = ((torch.rand(K, 10000, generator=g) > 0.9).float()[yi] - (torch.rand(N, 10000, generator=g) > 0.97).float()).abs()
dl = DataLoader(TensorDataset(X.to(dev), torch.eye(K)[yi].to(dev)), batch_size=B, shuffle=False)
params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95, "LEARNING_RATE_DYNAMIC": True,
          "CD": 1, "JOINT_LEARNING_RATE": 0.04, "JOINT_CD": 1, "JOINT_AUX_COND_STEPS": 30, "CROSS_GIBBS_STEPS": 50}
m = iMDBN([10000, 1500, 500], 256, params=params, dataloader=dl, val_loader=dl, device=dev, num_labels=K)
m.image_idbn.train(1)
import io, contextlib
for name, ep in (("first call (9 epochs: 8 warm-up + 1 main)", 9), ("again (9 epochs)", 9)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        m.train_joint(ep)
    torch.cuda.synchronize()
    print(f"train_joint {name}: {1e3 * (time.perf_counter() - t0) / (ep * N // B):.3f} ms per batch", flush=True)
