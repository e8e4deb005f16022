#!/usr/bin/env python3
"""Chain kernel (K4) probe: time of a 30-step noisy mean-field chain and a 50-step sampled Gibbs chain on the joint RBM
(532 <-> 256, one softmax group, batch 64 / 256) and the per-block timeline of chain step 2 (debug stamps)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import __graft_entry__ as ge
ge.build(compile_ok=False)
import torch
from imdbn import engine as E
from imdbn.engine import native
from imdbn.models import RBM
dev = torch.device("cuda")
eng = E.get_hip_engine()
for kv in sys.argv[1:]:
    k, v = kv.split("="); eng.set_option(k, int(v))
E.set_rng(E.PhiloxRng(seed=2))
for B in (64, 256):
    jr = RBM(532, 256, 0.04, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, softmax_groups=[(500, 532)]).to(dev)
    vk = torch.zeros(B, 532, device=dev); km = torch.zeros(B, 532, device=dev)
    vk[:, 500:] = torch.eye(32, device=dev)[torch.randint(0, 32, (B,), device=dev)]; km[:, 500:] = 1
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    nmf = t(lambda: jr.noisy_meanfield_annealed(vk, km, n_steps=30))
    gib = t(lambda: jr.conditional_gibbs(vk, km, n_steps=50, sample_h=True, sample_v=True))
    print(f"B={B}: noisy mean-field 30 steps {nmf:.0f} us ({nmf / 60:.2f} us per half step); sampled Gibbs 50 steps {gib:.0f} us ({gib / 100:.2f} us per half step)")
    for name, fn in (("noisy mean-field", lambda: jr.noisy_meanfield_annealed(vk, km, n_steps=30)),
                     ("sampled gibbs", lambda: jr.conditional_gibbs(vk, km, n_steps=50, sample_h=True, sample_v=True))):
        eng.set_option("dbg", 1024)
        fn(); torch.cuda.synchronize()
        buf = (C.c_longlong * (4096 * 8))()
        native.check(native.lib().imdbn_debug_stamps(buf, 4096 * 8), "imdbn_debug_stamps")
        eng.set_option("dbg", 0)
        a = np.frombuffer(buf, dtype=np.int64).reshape(4096, 8).copy()
        nb = int((a[:, 5] > 0).sum())
        a = a[:nb, :6].astype(np.float64) / 100.0
        d = np.diff(a, axis=1)
        print(f"   {name}: {nb} blocks; step 2 phases p50 (us): gemm h|v {np.median(d[:,0]):.2f}, epilogue h {np.median(d[:,1]):.2f}, gemm v|h {np.median(d[:,2]):.2f}, "
              f"epilogue v {np.median(d[:,3]):.2f}, group {np.median(d[:,4]):.2f}; whole step {np.median(a[:,5]-a[:,0]):.2f}")
