#!/usr/bin/env python3
"""Randomised parity stress: many random layer / batch shapes (aligned and unaligned rows, short and split-K visible
dimensions, one to four 64-row chunks, optional softmax group), each run through two consecutive CD-k updates (the
second one with its operand forms prefetched by the first), a mean-field chain and a clamped update, against the
numpy oracle in PHILOX mode, plus the data-parallel factor exchange in wire form with emulated ranks.  A mismatch whose
oracle run had a Bernoulli draw decided at rounding level (|p - u| < 3e-6) is reported as a near-tie flip, not a failure.
Prints one line per failure and a summary; exit code 1 on any failure.
    python tools/stress_parity.py [n_cases] [seed]"""
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
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)

def rel(a, b, atol):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), atol * np.sqrt(b.size) + 1e-30))

FAST = bool(os.environ.get("STRESS_FAST"))      # single-term bf16 weights (IMDBN_FAST_BF16): only finiteness can be checked
if FAST:
    from imdbn.engine import native as _native_mod
    E.get_hip_engine().mode = _native_mod.FAST_BF16
fails, ties, t0 = 0, 0, time.time()
TIE = 3e-6        # |p - u| below this: the sample is decided by rounding (fp32 sigmoid of a 1e-6-accurate pre-activation)
for case in range(n_cases):
    V = int(g.integers(9, 2600)); H = int(g.integers(4, 700)); B = int(g.integers(1, 230))
    if os.environ.get("STRESS_BIG"):          # the tile-height and multi-tile regimes of the large layers
        V = int(g.integers(4000, 9000)); H = int(g.integers(8, 1100)); B = int(g.integers(1, 140))
    if g.random() < 0.6:
        H = H // 4 * 4 + 4
    if g.random() < 0.3:
        V = V // 4 * 4 + 4
    groups = None
    if g.random() < 0.4 and V > 40:
        wd = int(g.integers(2, 40)); groups = [(V - wd, V)]
    cd = int(g.integers(1, 3)); binary = g.random() < 0.5
    mixed = (not binary) and g.random() < 0.5          # 0/1 pixels with a stretch of grey levels and a few exact halves: the per-item forms
    W0 = (g.standard_normal((V, H)) / np.sqrt(V)).astype(F32)
    hb = (g.standard_normal(H) * 0.1).astype(F32); vb = (g.standard_normal(V) * 0.1).astype(F32)
    kw = dict(dynamic_lr=True, final_momentum=0.95, softmax_groups=groups, sparsity=bool(g.random() < 0.3), sparsity_factor=0.1)
    r = RBM(V, H, 0.1, 1e-4, 0.5, **kw)
    P.set_params(r, DEV, W0, hb, vb)
    st = O.RBMState.create(W0, 0.1, 1e-4, 0.5, hid_bias=hb, vis_bias=vb, **kw)
    Xs = []
    for _ in range(2):
        X = g.random((B, V), dtype=F32)
        X = (X > 0.6).astype(F32) if binary else X
        if mixed:
            X = (X > 0.6).astype(F32)
            wdt = int(g.integers(1, max(2, V // 3))); a0 = int(g.integers(0, V - wdt + 1))
            X[:, a0:a0 + wdt] = g.random((B, wdt), dtype=F32)
            X[int(g.integers(0, B)), int(g.integers(0, V)):][:70] = 0.5
        Xs.append(X)
    tag = f"case {case}: V={V} H={H} B={B} groups={groups} cd={cd} binary={binary} mixed={mixed} sparsity={kw['sparsity']}"
    def run_once(seed):
        r = RBM(V, H, 0.1, 1e-4, 0.5, **kw)
        P.set_params(r, DEV, W0, hb, vb)
        st = O.RBMState.create(W0, 0.1, 1e-4, 0.5, hid_bias=hb, vis_bias=vb, **kw)
        # stale workspace contents must never matter: poison it (NaN) before the first call
        E.get_hip_engine()._workspace(torch.device(DEV), V, H, B).view(torch.float32).fill_(float("nan"))
        with E.use_rng(E.PhiloxRng(seed=seed)):
            d0, d1 = P.T(Xs[0], DEV), P.T(Xs[1], DEV)
            l0 = float(r.train_epoch(d0, 2, 10, CD=cd, next_data=d1))
            l1, fw = r.train_epoch(d1, 2, 10, CD=cd, return_forward=True)      # update + forward as one engine call
            l1, fw = float(l1), P.N(fw)
            Dz = groups[0][0] if groups else max(1, V // 3)
            vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
            vk[:, :Dz] = Xs[0][:, :Dz]; km[:, :Dz] = 1
            out = P.N(r.conditional_gibbs(P.T(vk, DEV), P.T(km, DEV), n_steps=3))
            lc = float(r.train_epoch_clamped(P.T(vk, DEV), P.T(km, DEV), 2, 10, CD=1, cond_init_steps=10, sample_h=False))
            # stand-alone propagations on the final weights: forward of the untagged / tagged batch == the fused forward, bit for bit;
            # backward (visible_probs) of a multi-chunk batch of real-valued rows against the oracle
            fa = r.forward(P.T(Xs[1], DEV))
            xt = P.T(Xs[1], DEV); xt._imdbn_binary = bool(binary)
            fb = r.forward(xt)
            hq = g.random((B, H), dtype=F32)
            bw = P.N(r.backward(P.T(hq, DEV)))
        O.reset_margin()
        s = PhiloxStream(seed)
        o0 = O.train_epoch(st, Xs[0], 2, cd, s)
        o1 = O.train_epoch(st, Xs[1], 2, cd, s)
        of = O.forward(st, Xs[1])
        oo = O.conditional_gibbs(st, vk, km, s, n_steps=3)
        oc = O.train_epoch_clamped(st, vk, km, 2, s, CD=1, cond_init_steps=10, sample_h=False)
        errs = {"loss0": abs(l0 - o0) / max(abs(o0), 1e-6), "loss1": abs(l1 - o1) / max(abs(o1), 1e-6),
                "chain": rel(out, oo, 1e-6), "forward": rel(fw, of, 1e-6), "lossc": abs(lc - oc) / max(abs(oc), 1e-4)}
        errs["backward"] = rel(bw, O.backward(st, hq), 1e-6)
        errs["forward (final weights)"] = rel(P.N(fa), O.forward(st, Xs[1]), 1e-6)
        errs["forward untagged vs tagged"] = 0.0 if torch.equal(fa, fb) else 1.0
        for k in P.KEYS:
            errs[k] = rel(P.N(getattr(r, k)), getattr(st, k), 2e-6)
        if FAST:
            fin = all(np.isfinite(P.N(getattr(r, k))).all() for k in P.KEYS) and np.isfinite([l0, l1, lc]).all() and np.isfinite(out).all()
            return ({} if fin else {"finite": 1.0}), float("inf")
        return {k: v for k, v in errs.items() if not (v < (2e-3 if k == "lossc" else 3e-4))}, O.BERNOULLI_MARGIN["min"]

    try:
        seed = int(g.integers(1, 1 << 30))
        bad, margin = run_once(seed)
        if bad and margin < TIE:
            # a sample decided at rounding level: one flip (and what it drags along) explains the difference -- but the
            # same shapes must then agree with other draws
            bad2, margin2 = run_once(seed + 1)
            if bad2 and margin2 >= TIE:
                fails += 1
                print("FAIL", tag, f"(retry after a near-tie) margin {margin2:.1e}", {k: f"{v:.2e}" for k, v in bad2.items()}, flush=True)
            else:
                ties += 1
                print("near-tie", tag, f"margin {margin:.1e}", {k: f"{v:.2e}" for k, v in bad.items()}, "-> retry", "near-tie again" if bad2 else "ok", flush=True)
        elif bad:
            fails += 1
            print("FAIL", tag, f"margin {margin:.1e}", {k: f"{v:.2e}" for k, v in bad.items()}, flush=True)
    except Exception as e:
        fails += 1
        print("ERROR", tag, type(e).__name__, str(e)[:200], flush=True)
    # data-parallel factor exchange in wire form, emulated ranks (eligible shapes only)
    try:
        eng = E.get_hip_engine()
        Bl = min(B, 64)
        if groups is None and H % 4 == 0 and eng.factor_mode_ok(r, Bl):
            R = int(g.integers(1, 5))
            kw2 = dict(kw); kw2["softmax_groups"] = None
            r2 = RBM(V, H, 0.1, 1e-4, 0.5, **kw2); P.set_params(r2, DEV, W0, hb, vb)
            st2 = O.RBMState.create(W0, 0.1, 1e-4, 0.5, hid_bias=hb, vis_bias=vb, **kw2)
            Xg = g.random((R * Bl, V), dtype=F32)
            Xg = (Xg > 0.6).astype(F32) if binary else Xg
            seed = int(g.integers(1, 1 << 30))
            lr_, mom_ = r2._lr_mom(2)
            if case % 2 == 0:
                wires = torch.stack([eng.pack_factors(r2, eng.cd_factors(r2, P.T(Xg[k * Bl:(k + 1) * Bl], DEV), cd, E.PhiloxRng(seed=seed, row0=k * Bl)), Bl, binary).clone()
                                     for k in range(R)])
                ld = float(eng.apply_factors_wire(r2, wires, eng.unpack_factors(r2, wires, Bl, binary, planes_only=True), Bl, R * Bl, lr_, mom_))
            else:       # the fused halves (cd_factors_wire with the next rank's rows as the prefetch hint, apply_wire)
                shards = [P.T(Xg[k * Bl:(k + 1) * Bl], DEV) for k in range(R)]
                wires = torch.stack([eng.cd_factors_wire(r2, shards[k], cd, E.PhiloxRng(seed=seed, row0=k * Bl), binary,
                                                         next_data=shards[k + 1] if k + 1 < R else None).clone() for k in range(R)])
                ld = float(eng.apply_wire(r2, wires, Bl, R * Bl, binary, lr_, mom_))
            O.reset_margin()
            od = O.train_epoch(st2, Xg, 2, cd, PhiloxStream(seed))
            errs = {"dp loss": abs(ld - od) / max(abs(od), 1e-6)}
            for k in P.KEYS:
                errs["dp " + k] = rel(P.N(getattr(r2, k)), getattr(st2, k), 2e-6)
            bad = {k: v for k, v in errs.items() if not (np.isfinite(v) if FAST else v < 3e-4)}
            if bad and O.BERNOULLI_MARGIN["min"] < TIE:
                ties += 1
                print("near-tie(dp)", tag, f"R={R} Bl={Bl} margin {O.BERNOULLI_MARGIN['min']:.1e}", {k: f"{v:.2e}" for k, v in bad.items()}, flush=True)
            elif bad:
                fails += 1
                print("FAIL(dp)", tag, f"R={R} Bl={Bl} margin {O.BERNOULLI_MARGIN['min']:.1e}", {k: f"{v:.2e}" for k, v in bad.items()}, flush=True)
    except Exception as e:
        fails += 1
        print("ERROR(dp)", tag, type(e).__name__, str(e)[:200], flush=True)
    # data-parallel all-reduce path (any shape: softmax groups, batches > 64 rows): two row shards' packed statistics,
    # summed, applied -- against the oracle's single update of the whole batch
    try:
        eng = E.get_hip_engine()
        if B >= 2:
            r3 = RBM(V, H, 0.1, 1e-4, 0.5, **kw); P.set_params(r3, DEV, W0, hb, vb)
            st3 = O.RBMState.create(W0, 0.1, 1e-4, 0.5, hid_bias=hb, vis_bias=vb, **kw)
            half = B // 2
            seed = int(g.integers(1, 1 << 30))
            s0 = eng.cd_stats(r3, P.T(Xs[0][:half], DEV), cd, E.PhiloxRng(seed=seed, row0=0)).clone()
            s1 = eng.cd_stats(r3, P.T(Xs[0][half:], DEV), cd, E.PhiloxRng(seed=seed, row0=half)).clone()
            lr_, mom_ = r3._lr_mom(2)
            la = float(eng.apply_delta(r3, s0 + s1, B, lr_, mom_))
            O.reset_margin()
            oa = O.train_epoch(st3, Xs[0], 2, cd, PhiloxStream(seed))
            errs = {"ar loss": abs(la - oa) / max(abs(oa), 1e-6)}
            for k in P.KEYS:
                errs["ar " + k] = rel(P.N(getattr(r3, k)), getattr(st3, k), 2e-6)
            bad = {k: v for k, v in errs.items() if not (np.isfinite(v) if FAST else v < 3e-4)}
            if bad and O.BERNOULLI_MARGIN["min"] < TIE:
                ties += 1
                print("near-tie(allreduce)", tag, f"margin {O.BERNOULLI_MARGIN['min']:.1e}", {k: f"{v:.2e}" for k, v in bad.items()}, flush=True)
            elif bad:
                fails += 1
                print("FAIL(allreduce)", tag, f"margin {O.BERNOULLI_MARGIN['min']:.1e}", {k: f"{v:.2e}" for k, v in bad.items()}, flush=True)
    except Exception as e:
        fails += 1
        print("ERROR(allreduce)", tag, type(e).__name__, str(e)[:200], flush=True)
    if case % 20 == 19:
        print(f"... {case + 1} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
print(f"stress_parity: {n_cases} cases, {fails} failures, {ties} near-tie sample flips, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
