#!/usr/bin/env python3
"""Per-block timeline of the 1500 -> 10000 decode launch (gemm_down_fused, real-valued hidden rows, batch B)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import torch
import __graft_entry__ as ge
ge.build()
from imdbn import engine as E
from imdbn.engine import native
from imdbn.models import RBM

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
V, H = 10000, 1500
dev = torch.device("cuda")
eng = E.get_hip_engine()
rbm = RBM(V, H, 0.1, 1e-4, 0.5).to(dev)
h = torch.rand(B, H, device=dev)
eng.set_option("dbg", 128)
for i in range(10):
    rbm.backward(h)
torch.cuda.synchronize()
buf = (C.c_longlong * (4096 * 8))()
native.check(native.lib().imdbn_debug_stamps(buf, 4096 * 8), "imdbn_debug_stamps")
a = np.frombuffer(buf, dtype=np.int64).reshape(4096, 8).copy()
nb = int((a[:, 0] > 0).sum())
a = a[:nb, :7].astype(np.float64) / 100.0
a -= a[:, 0].min()
print(f"{nb} blocks; slots: kernel start, body start, K loop done, barrier, reduced, epilogue done, operand tile flushed (us since the first block)")
for q, nm in ((0, "min"), (10, "p10"), (50, "p50"), (90, "p90"), (100, "max")):
    print(f"   {nm:4s} " + " ".join(f"{np.percentile(a[:, j], q):7.2f}" for j in range(7)))
d = np.diff(a, axis=1)
print("   phase durations p50: " + " ".join(f"{np.percentile(d[:, j], 50):7.2f}" for j in range(6)))
print("   phase durations p90: " + " ".join(f"{np.percentile(d[:, j], 90):7.2f}" for j in range(6)))
st = np.sort(a[:, 0]); print("   block starts: first 512 by", st[min(511, nb - 1)], " block 513..1024 by", st[min(1023, nb - 1)], " last", st[-1])
eng.set_option("dbg", 0)
