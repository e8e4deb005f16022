#!/usr/bin/env python3
"""Cost of the factor block's wire form: pack (one block) and unpack (R blocks), outside any collective."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import torch
import __graft_entry__ as ge
ge.build()
from imdbn import engine as E
from imdbn.models import RBM
dev = torch.device("cuda:0"); eng = E.get_hip_engine()
V, H, B = 10000, 1500, 64
rbm = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(dev)
x = (torch.rand(B, V) > 0.9).float().to(dev)
block = eng.cd_factors(rbm, x, 1, E.PhiloxRng(seed=1)).clone()
def timeit(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / n
for binary in (False, True):
    print(f"binary={binary}: full {block.numel()/1e6:.2f} MB -> wire {eng.compact_bytes(V, H, B, binary)/1e6:.2f} MB; pack {timeit(lambda: eng.pack_factors(rbm, block, B, binary)):.1f} us")
    for R in (1, 2, 4, 8):
        wires = torch.stack([eng.pack_factors(rbm, block, B, binary).clone() for _ in range(R)])
        print(f"   unpack {R} blocks: {timeit(lambda: eng.unpack_factors(rbm, wires, B, binary)):.1f} us")

# does it matter that the compact blocks were written by an RCCL all-gather?  (kernel time by HIP events)
import socket
import torch.distributed as dist
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
def ev_time(fn, prep, n=50):
    tot = 0.0
    for _ in range(n):
        prep()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return 1e3 * tot / n
wire = eng.pack_factors(rbm, block, B, True).clone()
out_rccl = torch.zeros(1, wire.numel(), dtype=torch.uint8, device=dev)
out_copy = torch.zeros(1, wire.numel(), dtype=torch.uint8, device=dev)
print("unpack after RCCL all_gather : %.1f us" % ev_time(lambda: eng.unpack_factors(rbm, out_rccl, B, True),
                                                          lambda: dist.all_gather_into_tensor(out_rccl.view(-1), wire)))
print("unpack after torch copy      : %.1f us" % ev_time(lambda: eng.unpack_factors(rbm, out_copy, B, True),
                                                          lambda: out_copy[0].copy_(wire)))
print("unpack, buffer untouched     : %.1f us" % ev_time(lambda: eng.unpack_factors(rbm, out_copy, B, True), lambda: None))
print("apply_factors after unpack   : %.1f us" % ev_time(lambda: eng.apply_factors(rbm, eng.gather_buffer(rbm, B, 1), B, B, 0.01, 0.5),
                                                          lambda: eng.unpack_factors(rbm, out_copy, B, True)))
dist.destroy_process_group()
