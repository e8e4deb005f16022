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
V, H, B = 10000, 1500, 64
xs = T._batches("binary", 4, B, V, seed=3)
def run(tag, prefetch, opts=()):
    for o in opts:
        k, v = o.split("="); eng.set_option(k, int(v))
    r, _ = T._rbm(V, H, 4)
    l = T._run(r, xs, tag, prefetch)
    torch.cuda.synchronize()
    for o in opts:
        k, v = o.split("="); eng.set_option(k, 0)
    return r, l
res = {}
for name, tag, pf, opts in (("untagged+pf", None, True, ()), ("tagged+pf", True, True, ()), ("untagged nopf", None, False, ()), ("tagged nopf", True, False, ()),
                            ("old path (no_k1s)", None, False, ("no_k1s=1",)), ("untagged+pf again", None, True, ())):
    r, l = run(tag, pf, opts)
    res[name] = (r, l)
    print(f"{name:22s} losses {[round(float(x), 7) for x in l]}  W sum {float(r.W.data.double().sum()):.6f}")
