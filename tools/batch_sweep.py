#!/usr/bin/env python3
"""CD-1 update time of the headline RBM (10000 <-> 1500) over the per-GPU batch size."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")          # a 16-cpu cgroup share: spinning pool threads throttle the enqueue thread
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import torch
import __graft_entry__ as ge
ge.build()
from imdbn import engine as E
from imdbn.models import RBM
dev = torch.device("cuda:0"); E.manual_seed(3)
rbm = RBM(10000, 1500, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(dev)
for B in (32, 64, 100, 128, 256, 512):
    x = (torch.rand(B, 10000) > 0.9).float().to(dev)
    for _ in range(5): rbm.train_epoch(x, 0, 1, CD=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): rbm.train_epoch(x, 0, 1, CD=1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(f"B={B:4d}: {1e6*dt:8.1f} us/update  ({B/dt/1e3:8.1f} k samples/s)")
