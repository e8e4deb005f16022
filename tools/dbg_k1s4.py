#!/usr/bin/env python3
"""Debug aid: which K rows explain the error of the negative-phase hidden probabilities of k1_stream?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from imdbn import engine as E
from imdbn.models import RBM
F32 = np.float32
dev = torch.device("cuda:0")
eng = E.get_hip_engine()
V, H, B = 10000, 1500, 64
g = np.random.Generator(np.random.PCG64(3))
W0 = (g.standard_normal((V, H), dtype=F32) * F32(0.01)).astype(F32)
X = (g.random((B, V), dtype=F32) > 0.9).astype(F32)
r = RBM(V, H, 0.1, 1e-4, 0.5).to(dev)
r.W.data.copy_(torch.from_numpy(W0).to(dev)); r.W_m.zero_()
eng.cd_factors(r, torch.from_numpy(X).to(dev), 1, E.PhiloxRng(seed=21), data_binary=True)
torch.cuda.synchronize()
Bp = 64
plane = eng.debug_buffer(dev, V, H, B, "vis_tr1", V * Bp * 2).cpu().numpy().view(np.uint16).reshape(V, Bp)
vneg = (plane != 0).astype(np.float64).T                       # [B][V]
ht = eng.debug_buffer(dev, V, H, B, "hid_tr1", 3 * H * Bp * 2).cpu().numpy().view(np.uint16).reshape(3, H, Bp)
def bf(x): return (x.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
pneg = -(bf(ht[0]) + bf(ht[1]) + bf(ht[2])).T                  # [B][H] (stored negated)
logit_gpu = np.log(pneg) - np.log1p(-pneg)
logit_ref = vneg @ W0.astype(np.float64)
d = logit_gpu - logit_ref
print("max |logit diff| per batch row:", np.round(np.abs(d).max(1), 5).tolist())
b = 0
# which K16 steps explain the difference for batch row 0?  contributions per step: c[j][n] = sum_{k in step j} v[k] W[k][n]
c = (vneg[b][:, None] * W0.astype(np.float64)).reshape(V // 16, 16, H).sum(1)      # [625][H]
# least squares: d[b] ~ sum_j a_j c[j]
a, *_ = np.linalg.lstsq(c.T, d[b], rcond=None)
idx = np.argwhere(np.abs(a) > 0.2).ravel()
print("steps with |coef| > 0.2:", [(int(i), round(float(a[i]), 3)) for i in idx][:60])
print("residual", np.abs(c.T @ a - d[b]).max())
