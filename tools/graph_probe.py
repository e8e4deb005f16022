#!/usr/bin/env python3
"""Eager calls against hipGraph replays (imdbn.engine.CapturedSteps) for launch-bound sequences: one CD-1 update of small RBMs,
the two-layer C2 stack iteration, the C3 joint step."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import __graft_entry__ as ge
ge.build(compile_ok=False)
import torch
from imdbn import engine as E
from imdbn.models import RBM
dev = torch.device("cuda")
E.set_rng(E.PhiloxRng(seed=2))

def timeit(fn, n=300, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

def rbm(V, H, **kw):
    return RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, **kw).to(dev)

for V, H, B in ((784, 256, 32), (1500, 500, 64), (532, 256, 64), (10000, 1500, 64)):
    r = rbm(V, H)
    x = (torch.rand(B, V, device=dev) > 0.8).float(); x._imdbn_binary = True
    eager = timeit(lambda: r.train_epoch(x, 0, 1, CD=1))
    step = E.CapturedSteps(lambda: r.train_epoch(x, 0, 1, CD=1))
    graph = timeit(step)
    print(f"CD-1 update {V}x{H} B={B}: eager {eager:.1f} us, graph replay {graph:.1f} us")

l1, l2 = rbm(10000, 1500), rbm(1500, 500)
x = (torch.rand(64, 10000, device=dev) > 0.9).float(); x._imdbn_binary = True
def c2():
    _, h = l1.train_epoch(x, 0, 1, CD=1, return_forward=True)
    return l2.train_epoch(h, 0, 1, CD=1)
eager = timeit(c2, 200)
graph = timeit(E.CapturedSteps(c2), 200)
print(f"C2 stack iteration (no next-batch prefetch): eager {eager:.1f} us, graph replay {graph:.1f} us")

jr = rbm(532, 256, softmax_groups=[(500, 532)])
z = torch.rand(64, 500, device=dev); y = torch.eye(32, device=dev)[torch.randint(0, 32, (64,), device=dev)]
vp = torch.cat([z, y], 1); vk = torch.zeros(64, 532, device=dev); km = torch.zeros(64, 532, device=dev)
vk[:, 500:] = y; km[:, 500:] = 1
def c3():
    jr.train_epoch(vp, 9, 20, CD=1)
    return jr.train_epoch_clamped(vk, km, 9, 20, CD=1, cond_init_steps=30, sample_h=False, sample_v=False, reclamp_negative=False,
                                  aux_lr_mult=0.3, use_noisy_init=True)
eager = timeit(c3, 100)
graph = timeit(E.CapturedSteps(c3), 100)
print(f"C3 joint main step: eager {eager:.1f} us, graph replay {graph:.1f} us")
