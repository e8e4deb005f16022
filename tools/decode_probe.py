#!/usr/bin/env python3
"""decode (500 -> 1500 -> 10000) and represent (10000 -> 1500 -> 500) of the image stack at batch B: wall time per call;
under `rocprofv3 --kernel-trace` the kernel sequence comes from tools/trace_seq.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import torch
import __graft_entry__ as ge
ge.build()
from imdbn.models import iDBN

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
which = sys.argv[2] if len(sys.argv) > 2 else "both"
dev = torch.device("cuda")
m = iDBN([10000, 1500, 500], {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95,
                              "LEARNING_RATE_DYNAMIC": True, "CD": 1}, None, None, device=dev)
z = torch.rand(B, 500, device=dev)
xb = (torch.rand(B, 10000, device=dev) > 0.9).float()
xr = torch.rand(B, 10000, device=dev)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
if which in ("both", "decode"):
    print(f"decode B={B}: {timeit(lambda: m.decode(z)):.1f} us")
if which in ("both", "represent"):
    print(f"represent (0/1 rows) B={B}: {timeit(lambda: m.represent(xb)):.1f} us")
    print(f"represent (real rows) B={B}: {timeit(lambda: m.represent(xr)):.1f} us")
