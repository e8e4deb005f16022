#!/usr/bin/env python3
"""Where does the HOST spend its time in iDBN.train at [10000, 1500, 500]?  (enqueue time vs device time per batch; cProfile)"""
import os, sys, time, tempfile, io, cProfile, pstats
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
X = (torch.rand(64 * 32, 10000, device=dev) > 0.9).float()
dl = DataLoader(TensorDataset(X, torch.zeros(len(X), 1, device=dev)), batch_size=64, shuffle=False)
params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95, "LEARNING_RATE_DYNAMIC": True, "CD": 1}
for ov in (False,):
    d = iDBN([10000, 1500, 500], dict(params), dl, dl, dev)
    d.train(1); torch.cuda.synchronize()
    t0 = time.perf_counter(); d.train(3); torch.cuda.synchronize()
    print(f"overlap={ov}: {1e6 * (time.perf_counter() - t0) / (3 * len(dl)):.1f} us per batch", flush=True)
    if "--prof" in sys.argv:
        pr = cProfile.Profile(); pr.enable(); d.train(3); torch.cuda.synchronize(); pr.disable()
        st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22); print(st.getvalue()[:5000])
