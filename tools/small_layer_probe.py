#!/usr/bin/env python3
"""CD-1 update time of a small layer (default 1500 <-> 500, batch 64, real-valued input as layer 2 of the stack sees it) under
different kernel choices (engine options given as name=value lists separated by '/')."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import __graft_entry__ as ge
ge.build(compile_ok=False)
import torch
from imdbn import engine as E
from imdbn.models import RBM
dev = torch.device("cuda")
eng = E.get_hip_engine()
V, H, B = (int(x) for x in os.environ.get("SHAPE", "1500,500,64").split(","))
variants = sys.argv[1:] or [""]
E.set_rng(E.PhiloxRng(seed=2))
for var in variants:
    opts = [kv.split("=") for kv in var.split("/") if kv]
    for k, v in opts: eng.set_option(k, int(v))
    r = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(dev)
    xs = [torch.rand(B, V, device=dev) for _ in range(8)]
    for x in xs: x._imdbn_binary = False
    def run(n):
        for i in range(n): r.train_epoch(xs[i % 8], 0, 1, CD=1)
    run(20); torch.cuda.synchronize(); t0 = time.perf_counter(); run(200); torch.cuda.synchronize()
    print(f"{V}x{H} B={B} [{var or 'default'}]: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per update")
    for k, v in opts: eng.set_option(k, 0)
