#!/usr/bin/env python3
"""Where does the HOST spend its time in iMDBN.train_joint?  (enqueue time vs device time per batch; cProfile of one call)"""
import os, sys, time, tempfile, io, contextlib, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import __graft_entry__ as ge
ge.build(compile_ok=False)
import torch
from torch.utils.data import DataLoader, TensorDataset
from imdbn import engine as E
from imdbn.models import iMDBN
dev = torch.device("cuda:0")
E.manual_seed(0)
os.chdir(tempfile.mkdtemp())
N, K, B = 64 * 20, 32, 64
g = torch.Generator().manual_seed(1)
yi = torch.randint(0, K, (N,), generator=g)
X = ((torch.rand(K, 10000, generator=g) > 0.9).float()[yi] - (torch.rand(N, 10000, generator=g) > 0.97).float()).abs()
dl = DataLoader(TensorDataset(X.to(dev), torch.eye(K)[yi].to(dev)), batch_size=B, shuffle=False)
for overlap in (True, False):
    params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95, "LEARNING_RATE_DYNAMIC": True,
              "CD": 1, "JOINT_LEARNING_RATE": 0.04, "JOINT_CD": 1, "JOINT_AUX_COND_STEPS": 30, "CROSS_GIBBS_STEPS": 50, "JOINT_METRICS_OVERLAP": overlap}
    m = iMDBN([10000, 1500, 500], 256, params=params, dataloader=dl, val_loader=dl, device=dev, num_labels=K)
    with contextlib.redirect_stdout(io.StringIO()):
        m.train_joint(1)
    torch.cuda.synchronize()
    for ep in (2, 9):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            m.train_joint(ep)
        torch.cuda.synchronize()
        print(f"overlap={overlap}: train_joint({ep}): {1e3 * (time.perf_counter() - t0) / (ep * N // B):.3f} ms per batch", flush=True)
pr = cProfile.Profile()
pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    m.train_joint(2)
torch.cuda.synchronize()
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(35)
print(st.getvalue()[:6000])
