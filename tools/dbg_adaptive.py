"""Debug aid: where do the item-wise (adaptive) prefetch and the in-call preparation first differ?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
from imdbn import engine as E
import parity_cases as P
import test_adaptive_gpu as T
eng = E.get_hip_engine()
opts = [o for o in sys.argv[1:] if "=" in o]
V, H, B = 2600, 500, 64
xs = T._batches("mixed", 24, B, V, seed=5)

def run(tag, prefetch, sync, options=()):
    for o in options:
        k, v = o.split("="); eng.set_option(k, int(v))
    r, _ = T._rbm(V, H, 1)
    ts = [P.T(x, "cuda:0") for x in xs]
    snaps = []
    with E.use_rng(E.PhiloxRng(seed=31)):
        for i, t in enumerate(ts):
            nxt = ts[i + 1] if (prefetch and i + 1 < len(ts)) else None
            if tag is not None:
                t._imdbn_binary = tag
                if nxt is not None: nxt._imdbn_binary = tag
            l = r.train_epoch(t, 0, 10, CD=1, next_data=nxt)
            if sync: torch.cuda.synchronize()
            snaps.append({k: (getattr(r, k).data if hasattr(getattr(r, k), "data") else getattr(r, k)).clone() for k in P.KEYS} | {"loss": l.clone()})
    torch.cuda.synchronize()
    for o in options:
        k, v = o.split("="); eng.set_option(k, 0)
    return snaps

def cmp(na, a, nb, b):
    for i, (sa, sb) in enumerate(zip(a, b)):
        msg = []
        for k in sa:
            d = (sa[k] != sb[k])
            if int(d.sum()):
                msg.append(f"{k}: {int(d.sum())} differ first {d.nonzero()[:2].tolist()} max {float((sa[k]-sb[k]).abs().max()):.2e}")
        if msg:
            x = xs[i]; nbm = ((x != 0) & (x != 1))
            print(f"{na} vs {nb}: first difference at step {i}: " + "; ".join(msg))
            print("   nonbinary items of this batch:", sorted(set((np.nonzero(nbm.any(0))[0] // 64).tolist())))
            d = (sa["W"] != sb["W"]).any(1).nonzero().flatten().tolist()
            print("   W rows that differ (as 64-row items):", sorted(set(r // 64 for r in d)), "count", len(d))
            inex = (torch.from_numpy(x).view(torch.int32) & 0xFFFF) != 0
            print("   inexact items:", sorted(set((np.nonzero(inex.numpy().any(0))[0] // 64).tolist())))
            return
    print(f"{na} vs {nb}: identical over {len(a)} steps")

A = run(None, True, False)
A2 = run(None, True, True)
A3 = run(None, True, True)
Bn = run(None, False, False)
C = run(False, True, False)
D = run(None, True, False, ["no_adaptive=1"])
cmp("A(untagged,prefetch)", A, "A2(same, synced)", A2)
cmp("A2", A2, "B", Bn)
cmp("A3", A3, "B", Bn)
cmp("A", A, "B(untagged,no prefetch)", Bn)
cmp("A", A, "C(tagged real)", C)
cmp("A", A, "D(no_adaptive)", D)
cmp("B", Bn, "C", C)
cmp("B", Bn, "D", D)
