#!/usr/bin/env python3
"""Where the host time of one RBM.train_epoch call goes (cProfile over 2000 calls of a small update)."""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import __graft_entry__ as ge
ge.build(compile_ok=False)
import torch
from imdbn import engine as E
from imdbn.models import RBM
dev = torch.device("cuda")
E.set_rng(E.PhiloxRng(seed=2))
V, H, B = (int(x) for x in os.environ.get("SHAPE", "784,256,32").split(","))
r = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(dev)
x = (torch.rand(B, V, device=dev) > 0.8).float(); x._imdbn_binary = True
for _ in range(50): r.train_epoch(x, 0, 1, CD=1)
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n): r.train_epoch(x, 0, 1, CD=1)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{V}x{H} B={B}: host enqueue {1e6 * (t1 - t0) / n:.1f} us per call, with the GPU drained {1e6 * (t2 - t0) / n:.1f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(n): r.train_epoch(x, 0, 1, CD=1)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:4500])
