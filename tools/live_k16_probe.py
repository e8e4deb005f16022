#!/usr/bin/env python3
"""Why does _cross_reconstruct with live best-of-16 take 2.3 ms in one bench run and 5-7 ms in another?"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd")]
import __graft_entry__ as ge
ge.build(compile_ok=False)
import torch
from torch.utils.data import DataLoader, TensorDataset
from imdbn import engine as E
from imdbn.models import iDBN, iMDBN
dev = torch.device("cuda:0")
E.manual_seed(3)
os.chdir(tempfile.mkdtemp())
params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95, "LEARNING_RATE_DYNAMIC": True, "CD": 1,
          "JOINT_LEARNING_RATE": 0.04, "JOINT_CD": 1, "JOINT_AUX_COND_STEPS": 30, "CROSS_GIBBS_STEPS": 50}
X = (torch.rand(64 * 8, 10000) > 0.9).float()
dl = DataLoader(TensorDataset(X, torch.zeros(len(X), 1)), batch_size=64)
def timeit(fn, n, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
def stats(tag):
    s = torch.cuda.memory_stats()
    print(f"   [{tag}] device mallocs {s.get('num_device_alloc', -1)} frees {s.get('num_device_free', -1)} reserved {s['reserved_bytes.all.current'] / 1e6:.0f} MB allocated {s['allocated_bytes.all.current'] / 1e6:.0f} MB", flush=True)
def run(tag):
    m = iMDBN([10000, 1500, 500], 256, params=dict(params), dataloader=dl, val_loader=dl, device=dev, num_labels=32)
    m.z_class_mean = torch.rand(32, 500, device=dev)
    z5 = torch.rand(256, 500, device=dev)
    y5 = torch.eye(32, device=dev)[torch.randint(0, 32, (256,), device=dev)]
    t5 = timeit(lambda: m._cross_reconstruct(z5, y5, steps=50), 10)
    m.live_best_of_k, m.best_of_k = True, 16
    stats(tag + " before live")
    t5k = timeit(lambda: m._cross_reconstruct(z5, y5, steps=50), 10)
    stats(tag + " after live")
    import cProfile, pstats, io
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    for _ in range(5): m._cross_reconstruct(z5, y5, steps=50)
    t_enq = (time.perf_counter() - t0) / 5
    torch.cuda.synchronize(); pr.disable()
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(8)
    print(f"   host enqueue {1e3 * t_enq:.3f} ms per call"); print("\n".join(st.getvalue().splitlines()[4:18]))
    print(f"{tag}: default {1e3 * t5:.3f} ms, live K=16 {1e3 * t5k:.3f} ms", flush=True)
    return m
run("fresh process")
from imdbn.models import RBM
if "--headline" in sys.argv:
    eng = E.get_hip_engine()
    rb = RBM(10000, 1500, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(dev)
    bt = [(torch.rand(64, 10000) > 0.9).float().to(dev) for _ in range(16)]
    E.set_rng(E.PhiloxRng(seed=2, row0=0))
    if "--profile" in sys.argv: eng.profile(True)
    for i in range(41): rb.train_epoch(bt[i % 16], 0, 1, CD=1, next_data=bt[(i + 1) % 16])
    torch.cuda.synchronize()
    if "--profile" in sys.argv: print("profile_read", eng.profile_read()); eng.profile(False)
    E.manual_seed(3)
    run("after the headline steps" + (" with event brackets" if "--profile" in sys.argv else ""))
jr = RBM(532, 256, 0.04, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, softmax_groups=[(500, 532)]).to(dev)
z = torch.rand(64, 500, device=dev); y = torch.eye(32, device=dev)[torch.randint(0, 32, (64,), device=dev)]
vp = torch.cat([z, y], 1); vk = torch.zeros(64, 532, device=dev); km = torch.zeros(64, 532, device=dev); vk[:, 500:] = y; km[:, 500:] = 1
def c3_main():
    jr.train_epoch(vp, 9, 20, CD=1)
    jr.train_epoch_clamped(vk, km, 9, 20, CD=1, cond_init_steps=30, sample_h=False, sample_v=False, reclamp_negative=False, aux_lr_mult=0.3, use_noisy_init=True)
print("C3 main step", 1e3 * timeit(c3_main, 20), "ms")
run("after the C3 steps")
X2 = (torch.rand(64 * 32, 10000, device=dev) > 0.9).float()
dl2 = DataLoader(TensorDataset(X2, torch.zeros(len(X2), 1, device=dev)), batch_size=64, shuffle=False)
d = iDBN([10000, 1500, 500], dict(params), dl2, dl2, dev)
d.train(2)
torch.cuda.synchronize()
run("after the stack loop over a device-resident dataset")
run("again")
