#!/usr/bin/env python3
"""Randomised parity stress of the chains (joint-RBM shapes: the row-parallel chain kernel K4, and the per-launch path
for shapes it does not take): conditional_gibbs (mean-field / sampled), conditional_gibbs_annealed,
noisy_meanfield_annealed with and without the mu-pull, train_epoch_clamped, against the numpy oracle in PHILOX mode.
Mismatches with a Bernoulli draw decided at rounding level are re-run with other draws (see stress_parity.py).
    python tools/stress_chains.py [n_cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-idbn_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import __graft_entry__ as ge
ge.build()
import oracle.rbm_oracle as O
from oracle.draws import PhiloxStream
from imdbn import engine as E
from imdbn.models import RBM
import parity_cases as P

F32 = np.float32
DEV = "cuda:0"
TIE = 3e-6
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)

def rel(a, b, atol=2e-6):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), atol * np.sqrt(b.size) + 1e-30))

fails, ties, t0 = 0, 0, time.time()
for case in range(n_cases):
    big = g.random() < 0.2                                  # beyond the chain kernel's shapes: per-launch path
    V = int(g.integers(1030, 1800)) if big else int(g.integers(40, 1024))
    H = int(g.integers(8, 1024)); B = int(g.integers(1, 300))
    if g.random() < 0.6:
        H = H // 4 * 4 + 4
    wd = min(int(g.integers(2, 65)), V - 8) if g.random() < 0.8 else 0
    groups = [(V - wd, V)] if wd else None
    Dz = V - wd if wd else max(1, V // 2)
    W0 = (g.standard_normal((V, H)) / np.sqrt(V) * 2.0).astype(F32)
    hb = (g.standard_normal(H) * 0.2).astype(F32); vb = (g.standard_normal(V) * 0.2).astype(F32)
    kw = dict(dynamic_lr=True, final_momentum=0.95, softmax_groups=groups)
    vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
    if g.random() < 0.5 and wd:                              # labels known (TXT -> IMG), else the features known
        vk[np.arange(B), Dz + g.integers(0, wd, B)] = 1; km[:, Dz:] = 1
    else:
        vk[:, :Dz] = g.random((B, Dz), dtype=F32); km[:, :Dz] = 1
    kind = int(g.integers(0, 5))
    n = int(g.integers(1, 13))
    sh, sv = bool(g.random() < 0.5), bool(g.random() < 0.5)
    mu = g.random((B, Dz), dtype=F32) if g.random() < 0.5 else None
    tag = f"case {case}: V={V} H={H} B={B} group={wd} kind={kind} n={n} sample_h={sh} sample_v={sv} mu={mu is not None}"

    def run_once(seed):
        r = RBM(V, H, 0.1, 1e-4, 0.5, **kw); P.set_params(r, DEV, W0, hb, vb)
        st = O.RBMState.create(W0, 0.1, 1e-4, 0.5, hid_bias=hb, vis_bias=vb, **kw)
        if mu is not None:
            r._mu_pull = {"mu_k": P.T(mu, DEV), "eta0": 0.2}; st.mu_pull = {"mu_k": mu, "eta0": 0.2}
        a, b = P.T(vk, DEV), P.T(km, DEV)
        E.get_hip_engine()._workspace(torch.device(DEV), V, H, B).view(torch.float32).fill_(float("nan"))   # stale contents must not matter
        O.reset_margin()
        s = PhiloxStream(seed)
        with E.use_rng(E.PhiloxRng(seed=seed)):
            if kind == 0:
                out = P.N(r.conditional_gibbs(a, b, n_steps=n, sample_h=sh, sample_v=sv)); ref = O.conditional_gibbs(st, vk, km, s, n_steps=n, sample_h=sh, sample_v=sv)
            elif kind == 1:
                out = P.N(r.noisy_meanfield_annealed(a, b, n_steps=n)); ref = O.noisy_meanfield_annealed(st, vk, km, s, n_steps=n)
            elif kind == 2:
                k2 = dict(n_steps=n, sample_h_until=n // 2, sample_v_every=2 if sv else 0, final_meanfield=sh)
                out = P.N(r.conditional_gibbs_annealed(a, b, **k2)); ref = O.conditional_gibbs_annealed(st, vk, km, s, **k2)
            else:
                k3 = dict(CD=int(g2.integers(1, 3)), cond_init_steps=n, sample_h=sh, sample_v=sv, reclamp_negative=bool(kind == 3), use_noisy_init=bool(n % 2))
                l = float(r.train_epoch_clamped(a, b, 1, 10, **k3)); lo = O.train_epoch_clamped(st, vk, km, 1, s, **k3)
                errs = {"loss": abs(l - lo) / max(abs(lo), 1e-4)}
                for k in P.KEYS:
                    errs[k] = rel(P.N(getattr(r, k)), getattr(st, k))
                    if np.abs(P.N(getattr(r, k)).astype(np.float64) - np.asarray(getattr(st, k), np.float64)).max() < 2e-6:
                        errs[k] = 0.0                             # the tests' absolute tolerance (assert_close atol): cancellation noise
                    if k.endswith("_m") and errs[k] >= 1e-3:      # context for the report: absolute sizes
                        x, y = P.N(getattr(r, k)).astype(np.float64), np.asarray(getattr(st, k), np.float64)
                        print(f"      {k}: |ref| = {np.linalg.norm(y):.3e} (rms {np.sqrt((y ** 2).mean()):.2e}), |diff| = {np.linalg.norm(x - y):.3e}, max|diff| = {np.abs(x - y).max():.2e}", flush=True)
                # the momenta of a mean-field clamped update are differences of nearly equal sums: rounding is amplified
                lim = lambda k: 3e-3 if k == "loss" else (1e-3 if k.endswith("_m") else 3e-4)
                return {k: v for k, v in errs.items() if not v < lim(k)}, O.BERNOULLI_MARGIN["min"]
        e = rel(out, ref)
        return ({"chain": e} if not e < 3e-4 else {}), O.BERNOULLI_MARGIN["min"]

    try:
        seed = int(g.integers(1, 1 << 30))
        g2 = np.random.default_rng(seed)
        bad, margin = run_once(seed)
        if bad and margin < TIE:
            g2 = np.random.default_rng(seed)
            bad2, margin2 = run_once(seed + 1)
            if bad2 and margin2 >= TIE:
                fails += 1; print("FAIL", tag, f"(retry) margin {margin2:.1e}", {k: f"{v:.2e}" for k, v in bad2.items()}, flush=True)
            else:
                ties += 1; print("near-tie", tag, f"margin {margin:.1e}", {k: f"{v:.2e}" for k, v in bad.items()}, "-> retry", "near-tie again" if bad2 else "ok", flush=True)
        elif bad:
            fails += 1; print("FAIL", tag, f"margin {margin:.1e}", {k: f"{v:.2e}" for k, v in bad.items()}, flush=True)
    except Exception as e:
        fails += 1; print("ERROR", tag, type(e).__name__, str(e)[:300], flush=True)
    # the two chains of _cross_reconstruct in one launch (imdbn_rbm_chain_pair) == the two calls, bit for bit (any shape: shapes the
    # chain kernel does not take run the calls one after the other inside the library)
    try:
        if wd:
            r = RBM(V, H, 0.1, 1e-4, 0.5, **kw); P.set_params(r, DEV, W0, hb, vb)
            vz = np.zeros((B, V), F32); kz = np.zeros((B, V), F32)
            vz[:, :Dz] = g.random((B, Dz), dtype=F32); kz[:, :Dz] = 1
            vy = np.zeros((B, V), F32); ky = np.zeros((B, V), F32)
            vy[np.arange(B), Dz + g.integers(0, wd, B)] = 1; ky[:, Dz:] = 1
            gib = dict(v_known=P.T(vz, DEV), known_mask=P.T(kz, DEV), n_steps=n, sample_h=sh, sample_v=sv)
            nmf = dict(v_known=P.T(vy, DEV), known_mask=P.T(ky, DEV), n_steps=int(g.integers(1, 13)), T0=3.0, T1=1.0, sigma0=0.9, hot_frac=0.7,
                       sharpen_last=3, T_cold_plus=0.9)
            pull = {"mu_k": P.T(g.random((B, Dz), dtype=F32), DEV), "eta0": 0.15} if g.random() < 0.7 else None
            seed = int(g.integers(1, 1 << 30))
            with E.use_rng(E.PhiloxRng(seed=seed)) as rng:
                r._mu_pull = pull
                a1, b1 = r._chain_pair(gib, nmf)
                used = rng.offset
            with E.use_rng(E.PhiloxRng(seed=seed)) as rng:
                r._mu_pull = None
                a2 = r.conditional_gibbs(**gib)
                r._mu_pull = pull
                b2 = r.noisy_meanfield_annealed(**nmf)
                used2 = rng.offset
            if not (used == used2 and torch.equal(a1, a2) and torch.equal(b1, b2) and bool(torch.isfinite(a1).all()) and bool(torch.isfinite(b1).all())):
                fails += 1; print("FAIL (chain pair vs two calls)", tag, f"draws {used} vs {used2}; gibbs differs {int((a1 != a2).sum())}, mean-field differs {int((b1 != b2).sum())}", flush=True)
    except Exception as e:
        fails += 1; print("ERROR (chain pair)", tag, type(e).__name__, str(e)[:300], flush=True)
    if case % 25 == 24:
        print(f"... {case + 1} cases, {fails} failures, {ties} near-ties, {time.time() - t0:.0f} s", flush=True)
print(f"stress_chains: {n_cases} cases, {fails} failures, {ties} near-tie sample flips, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
