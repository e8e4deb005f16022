"""GPU parity of the objects and sizes the benchmark really runs.

* the RBM exactly as ``RBM(10000, 1500)`` constructs it on the GPU: W / W_m are views of row-padded buffers
  (pitch 1536 != H); the padding columns are poisoned with NaN, the fixture weights are copied INTO the view, and
  after the replayed digest updates and a PHILOX update against the oracle the padding must still be NaN-only and
  no NaN may have leaked into the model;
* BASELINE configs[2] (joint RBM 532 <-> 256, 30 auxiliary clamped steps), configs[3] (global batch 512 = 8 ranks x 64
  rows, both exchanges, emulated on one device) and configs[4] (_cross_reconstruct, batch 256, 50 steps, best-of-K)
  at their FULL sizes against the numpy oracle.
"""
import numpy as np
import pytest
import torch

import oracle.rbm_oracle as O
import parity_cases as P
from golden_utils import Fixture, assert_close, gpu_cd_samples, init_W, rel_fro
from oracle.draws import DrawStream, PhiloxStream

pytestmark = pytest.mark.gpu
F32 = np.float32
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _native():
    import __graft_entry__ as ge
    ge.build()
    from imdbn import engine as E
    E.set_engine_for_testing(None)
    yield E.get_hip_engine()


def _padded(r):
    """(W buffer incl. padding, W_m buffer incl. padding, pitch) of an RBM built by the constructor on the GPU."""
    V, H = r.W.shape
    pitch = r.W.stride(0)
    return (torch.as_strided(r.W.data, (V, pitch), (pitch, 1)), torch.as_strided(r.W_m, (V, pitch), (pitch, 1)), pitch)


def _poison_and_fill(r, W0):
    Wb, Mb, pitch = _padded(r)
    H = r.W.shape[1]
    assert r.W_m.stride(0) == pitch and r.W_m.data_ptr() != r.W.data_ptr()
    if pitch > H:
        Wb[:, H:] = float("nan")
        Mb[:, H:] = float("nan")
    r.W.data.copy_(P.T(W0, DEV))          # IN PLACE: the row-padded layout stays
    r.W_m.zero_()
    r.hid_bias.data.zero_(); r.vis_bias.data.zero_(); r.hb_m.zero_(); r.vb_m.zero_()
    assert r.W.stride(0) == pitch


def _padding_untouched(r):
    Wb, Mb, pitch = _padded(r)
    H = r.W.shape[1]
    if pitch > H:
        assert torch.isnan(Wb[:, H:]).all() and torch.isnan(Mb[:, H:]).all(), "a kernel wrote into the row padding"
    for k in P.KEYS:
        assert torch.isfinite(getattr(r, k)).all(), f"NaN leaked from the row padding into {k}"


@pytest.mark.parametrize("pitch_env,want_pitch", [(None, 1536), ("32", 1504)])
def test_headline_rbm_as_constructed_with_poisoned_padding(pitch_env, want_pitch, monkeypatch):
    """What bench.py times: RBM(10000, 1500) from the constructor (pitch 1536; 1504 with IMDBN_ROW_PITCH=32)."""
    from imdbn import engine as E
    from imdbn.models import RBM
    if pitch_env:
        monkeypatch.setenv("IMDBN_ROW_PITCH", pitch_env)
    fx = Fixture("c2_rbm10000x1500_cd1_digest.npz")
    m = fx.meta
    s = fx.stream()
    Vv, Hh, B, U = m["V"], m["H"], m["B"], m["updates"]
    r = RBM(Vv, Hh, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(DEV)
    assert r.W.stride(0) == want_pitch and r.W.stride(0) > Hh and not r.W.is_contiguous()
    W0 = init_W(s, Vv, Hh)
    _poison_and_fill(r, W0)
    X = (s.uniform((B * U, Vv)) > 0.9).astype(F32)
    with E.use_rng(E.ReplayRng(s)):
        losses = [float(r.train_epoch(P.T(X[B * i:B * i + B], DEV), 0, 10, CD=1)) for i in range(U)]
    _padding_untouched(r)
    assert_close(np.array(losses, F32), fx["losses"], 2e-5, "losses")
    for k in ("W", "W_m"):
        a = P.N(getattr(r, k))
        assert abs(a.astype(np.float64).sum() - fx[k + "_sum"]) <= 1e-4 * abs(fx[k + "_sum"]) + 1e-2
        assert abs((a.astype(np.float64) ** 2).sum() - fx[k + "_sumsq"]) <= 1e-4 * fx[k + "_sumsq"]
        assert_close(a.ravel()[fx[k + "_probe_idx"]], fx[k + "_probe_val"], 1e-4, k + " probes", atol=1e-6)
    for k in ("hid_bias", "vis_bias", "hb_m", "vb_m"):
        assert_close(P.N(getattr(r, k)), fx[k], 2e-4, k, atol=1e-6)

    # a PHILOX update (device draws) with the next-batch prefetch, against the oracle fed by the numpy Philox twin
    g = np.random.Generator(np.random.PCG64(3))
    Wn = (g.standard_normal((Vv, Hh), dtype=F32) * F32(0.01)).astype(F32)
    _poison_and_fill(r, Wn)
    Xa = (g.random((B, Vv), dtype=F32) > 0.9).astype(F32)
    Xb = (g.random((B, Vv), dtype=F32) > 0.9).astype(F32)
    ta, tb = P.T(Xa, DEV), P.T(Xb, DEV)
    eng = E.get_hip_engine()
    st = O.RBMState.create(Wn, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95)
    ps = PhiloxStream(21)
    with E.use_rng(E.PhiloxRng(seed=21)):
        # ~830 000 Bernoulli comparisons per update: a few |p - u| are at fp32 rounding level, where the summation order decides
        # (SURVEY 7.3-a).  The oracle takes the sample the device drew wherever its own margin is < 2e-6 and reports how often.
        for x, t, nxt in ((Xa, ta, tb), (Xb, tb, None)):
            l = r.train_epoch(t, 0, 10, CD=1, next_data=nxt)
            h_gpu, v_gpu = gpu_cd_samples(eng, DEV, Vv, Hh, B)
            O.set_tie_break([h_gpu, v_gpu, None], tol=2e-6)
            o = O.train_epoch(st, x, 0, 1, ps)
            assert O.TIE_BREAK["ties"] < 60 and O.TIE_BREAK["used"] <= 6, dict(O.TIE_BREAK, queue=None)
            O.set_tie_break(None)
            assert_close(float(l), o, 1e-5, "loss")
            for k in P.KEYS:
                assert_close(P.N(getattr(r, k)), getattr(st, k), 1e-4, k, atol=2e-6)
    _padding_untouched(r)


def test_headline_one_epoch_of_100_updates_against_the_oracle_reporting_flips():
    """north_star: "weights after one epoch" -- BASELINE configs[1] layer 1 for a whole epoch of 100 batch-64 CD-1 updates on the
    object bench.py times (constructor layout, next-batch prefetch, device Philox draws) against the numpy oracle fed by the
    Philox twin.  ~83 million Bernoulli comparisons: those whose oracle margin |p - u| is at rounding level are decided by the
    fp32 summation order, so the oracle takes the device's sample there -- and the test REPORTS how many such
    near-ties there were and how many the device decided the other way, instead of failing blind or skipping the comparison."""
    from imdbn import engine as E
    from imdbn.models import RBM
    Vv, Hh, B, U = 10000, 1500, 64, 100
    g = np.random.Generator(np.random.PCG64(11))
    W0 = (g.standard_normal((Vv, Hh), dtype=F32) * F32(0.01)).astype(F32)
    r = RBM(Vv, Hh, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(DEV)
    _poison_and_fill(r, W0)
    st = O.RBMState.create(W0, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95)
    proto = g.random((16, Vv), dtype=F32) < 0.1                                   # 16 "stimulus classes" with noise: not pure noise images
    Xs = [np.logical_xor(proto[g.integers(0, 16, B)], g.random((B, Vv), dtype=F32) < 0.02).astype(F32) for _ in range(U)]
    ts = [P.T(x, DEV) for x in Xs]
    eng = E.get_hip_engine()
    ps = PhiloxStream(5)
    ties = used = 0
    worst = 0.0
    with E.use_rng(E.PhiloxRng(seed=5)):
        for i in range(U):
            l = r.train_epoch(ts[i], 0, 1, CD=1, next_data=ts[i + 1] if i + 1 < U else None)
            h_gpu, v_gpu = gpu_cd_samples(eng, DEV, Vv, Hh, B)
            O.set_tie_break([h_gpu, v_gpu, None], tol=1e-5)
            o = O.train_epoch(st, Xs[i], 0, 1, ps)
            ties += O.TIE_BREAK["ties"]; used += O.TIE_BREAK["used"]
            O.set_tie_break(None)
            worst = max(worst, abs(float(l) - float(o)) / max(abs(float(o)), 1e-12))
    print(f"[one epoch] 100 updates: {ties} near-ties (|p-u| < 1e-5) of {U * B * (2 * Hh + Vv)} comparisons, {used} decided differently by the device; "
          f"worst relative loss difference {worst:.2e}")
    assert worst < 2e-5
    assert ties < 100 * U and used <= ties // 4 + 10, (ties, used)       # near-ties are ~1e-5 of the comparisons; the device flips a minority of them
    _padding_untouched(r)
    for k in P.KEYS:
        assert_close(P.N(getattr(r, k)), getattr(st, k), 1e-4, k, atol=2e-6)
    assert rel_fro(P.N(r.W), st.W) < 1e-5


def test_gpu_pickle_of_a_padded_rbm_is_contiguous(tmp_path):
    """A model saved from the GPU must not carry the padded storage: W unpickles as a plain contiguous [V, H] tensor
    (``W.is_contiguous()``, stride (H, 1)) with the same values."""
    import pickle
    from imdbn.models import RBM
    r = RBM(2000, 500, 0.1, 1e-4, 0.5).to(DEV)
    assert r.W.stride(0) == 512
    blob = pickle.dumps(r)
    assert len(blob) < 2 * 2000 * 500 * 4 + 200000, "the pickle carries the padded buffers"
    q = pickle.loads(blob)
    assert q.W.is_contiguous() and q.W.stride() == (500, 1) and q.W_m.is_contiguous()
    assert torch.equal(q.W.data.cpu(), r.W.data.cpu())


def test_config3_joint_rbm_full_size_against_oracle():
    """BASELINE configs[2]: joint RBM 532 <-> 256 with 32 softmax labels, batch 64: one CD-1 update and one auxiliary
    clamped update with 30 noisy mean-field initialisation steps (imdbn.py:590-611), PHILOX draws, vs the oracle."""
    from imdbn import engine as E
    from imdbn.models import RBM
    V, H, Dz, B = 532, 256, 500, 64
    g = np.random.Generator(np.random.PCG64(31))
    W0 = (g.standard_normal((V, H), dtype=F32) / F32(np.sqrt(V))).astype(F32)
    hb = (g.standard_normal(H, dtype=F32) * F32(0.1)).astype(F32)
    vb = (g.standard_normal(V, dtype=F32) * F32(0.1)).astype(F32)
    r = RBM(V, H, 0.04, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, softmax_groups=[(Dz, V)])
    P.set_params(r, DEV, W0, hb, vb)
    st = O.RBMState.create(W0, 0.04, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, softmax_groups=[(Dz, V)], hid_bias=hb, vis_bias=vb)
    z = g.random((B, Dz), dtype=F32)
    y = np.eye(V - Dz, dtype=F32)[g.integers(0, V - Dz, B)]
    vp = np.concatenate([z, y], 1)
    vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
    vk[:, Dz:] = y; km[:, Dz:] = 1
    O.reset_margin()
    with E.use_rng(E.PhiloxRng(seed=17)):
        l1 = r.train_epoch(P.T(vp, DEV), 9, 20, CD=1)
        l2 = r.train_epoch_clamped(P.T(vk, DEV), P.T(km, DEV), 9, 20, CD=1, cond_init_steps=30, sample_h=False, sample_v=False,
                                   reclamp_negative=False, aux_lr_mult=0.3, use_noisy_init=True)
    ps = PhiloxStream(17)
    o1 = O.train_epoch(st, vp, 9, 1, ps)
    o2 = O.train_epoch_clamped(st, vk, km, 9, ps, CD=1, cond_init_steps=30, sample_h=False, sample_v=False, reclamp_negative=False,
                               aux_lr_mult=0.3, use_noisy_init=True)
    assert_close(np.array([float(l1), float(l2)], F32), np.array([o1, o2], F32), 1e-4, "losses")
    for k in P.KEYS:
        assert_close(P.N(getattr(r, k)), getattr(st, k), 1e-4, k, atol=2e-6)


def test_config4_eight_ranks_of_64_rows_full_size(_native):
    """BASELINE configs[3]: 10000 <-> 1500, global batch 512 = 8 ranks x 64 rows (PHILOX keyed on the global row), emulated on
    one device: factor exchange (wire form, data declared binary), all-reduce of the packed statistics, the single-process
    8-chunk update of the same 512 rows and the oracle must all agree."""
    from imdbn import engine as E
    from imdbn.models import RBM
    V, H, R, Bl = 10000, 1500, 8, 64
    B = R * Bl
    g = np.random.Generator(np.random.PCG64(41))
    W0 = (g.standard_normal((V, H), dtype=F32) * F32(0.01)).astype(F32)
    X = (g.random((B, V), dtype=F32) > 0.9).astype(F32)

    def fresh():
        r = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(DEV)      # constructor layout (padded rows)
        _poison_and_fill(r, W0)
        return r
    eng = _native
    r1, r2, r3 = fresh(), fresh(), fresh()
    lr, mom = r1._lr_mom(0)
    xs = [P.T(X[rk * Bl:(rk + 1) * Bl], DEV) for rk in range(R)]
    with E.use_rng(E.PhiloxRng(seed=77)):
        l1 = r1.train_epoch(P.T(X, DEV), 0, 1, CD=1)                          # one rank, eight 64-row chunks
    dec1 = gpu_cd_samples(eng, DEV, V, H, B)
    assert eng.factor_mode_ok(r2, Bl)
    wires = eng.compact_gather_buffer(r2, Bl, R, True)
    dec2 = []
    for rk in range(R):
        blk = eng.cd_factors(r2, xs[rk], 1, E.PhiloxRng(seed=77, row0=rk * Bl))
        dec2.append(gpu_cd_samples(eng, DEV, V, H, Bl))
        wires[rk].copy_(eng.pack_factors(r2, blk, Bl, True))
    dec2 = (np.concatenate([d[0] for d in dec2]), np.concatenate([d[1] for d in dec2]))
    planes = eng.unpack_factors(r2, wires, Bl, True, planes_only=True)
    l2 = eng.apply_factors_wire(r2, wires, planes, Bl, B, lr, mom)
    packed = None
    for rk in range(R):
        s = eng.cd_stats(r3, xs[rk], 1, E.PhiloxRng(seed=77, row0=rk * Bl)).clone()
        packed = s if packed is None else packed + s
    l3 = eng.apply_delta(r3, packed, B, lr, mom)
    for r in (r1, r2, r3):
        _padding_untouched(r)
    # both exchanges run the same per-rank kernels: identical samples, identical update
    assert_close(float(l3), float(l2), 1e-6, "loss: all-reduce vs factor exchange")
    for k in P.KEYS:
        assert_close(P.N(getattr(r3, k)), P.N(getattr(r2, k)), 1e-5, "all-reduce vs factors: " + k, atol=2e-6)
    # 6.6 million Bernoulli comparisons: the 8-chunk launch and the 64-row launches split K differently, so a handful of
    # near-tie samples (|p - u| < 2e-6) may differ between them; each is checked against the oracle with ITS samples at the ties
    same = all(np.array_equal(a, b) for a, b in zip(dec1, dec2))
    if same:
        for k in P.KEYS:
            assert_close(P.N(getattr(r2, k)), P.N(getattr(r1, k)), 1e-5, "factors vs single process: " + k, atol=2e-6)
    for name, rr, ll, dec in (("single process", r1, l1, dec1), ("8 ranks", r2, l2, dec2)):
        st = O.RBMState.create(W0, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95)
        O.set_tie_break([dec[0], dec[1], None], tol=2e-6)
        o = O.train_epoch(st, X, 0, 1, PhiloxStream(77))
        assert O.TIE_BREAK["ties"] < 300 and O.TIE_BREAK["used"] <= 30, dict(O.TIE_BREAK, queue=None)
        O.set_tie_break(None)
        assert_close(float(ll), o, 1e-5, name + ": loss vs oracle")
        for k in P.KEYS:
            assert_close(P.N(getattr(rr, k)), getattr(st, k), 1e-4, name + " vs oracle: " + k, atol=2e-6)


def test_config4_second_layer_eight_ranks_real_valued_data(_native):
    """BASELINE configs[3], second layer of the stack: 1500 <-> 500 on REAL-valued data (first-layer probabilities), global batch
    512 = 8 ranks x 64 rows emulated on one device: factor exchange (fp32 wire form), all-reduce of the packed statistics, the
    single-process 8-chunk update of the same rows and the oracle must all agree."""
    from imdbn import engine as E
    from imdbn.models import RBM
    V, H, R, Bl = 1500, 500, 8, 64
    B = R * Bl
    g = np.random.Generator(np.random.PCG64(43))
    W0 = (g.standard_normal((V, H), dtype=F32) * F32(0.03)).astype(F32)
    X = g.random((B, V), dtype=F32)

    def fresh():
        r = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(DEV)      # constructor layout (padded rows)
        _poison_and_fill(r, W0)
        return r
    eng = _native
    r1, r2, r3 = fresh(), fresh(), fresh()
    lr, mom = r1._lr_mom(0)
    xs = [P.T(X[rk * Bl:(rk + 1) * Bl], DEV) for rk in range(R)]
    with E.use_rng(E.PhiloxRng(seed=77)):
        l1 = r1.train_epoch(P.T(X, DEV), 0, 1, CD=1)                          # one rank, eight 64-row chunks
    dec1 = gpu_cd_samples(eng, DEV, V, H, B)
    assert eng.factor_mode_ok(r2, Bl)
    wires = eng.compact_gather_buffer(r2, Bl, R, False)
    dec2 = []
    for rk in range(R):                                                         # the production calls of RBM.train_epoch under dp
        wire = eng.cd_factors_wire(r2, xs[rk], 1, E.PhiloxRng(seed=77, row0=rk * Bl), False)
        dec2.append(gpu_cd_samples(eng, DEV, V, H, Bl))
        wires[rk].copy_(wire)
    dec2 = (np.concatenate([d[0] for d in dec2]), np.concatenate([d[1] for d in dec2]))
    l2 = eng.apply_wire(r2, wires, Bl, B, False, lr, mom)
    packed = None
    for rk in range(R):
        s = eng.cd_stats(r3, xs[rk], 1, E.PhiloxRng(seed=77, row0=rk * Bl)).clone()
        packed = s if packed is None else packed + s
    l3 = eng.apply_delta(r3, packed, B, lr, mom)
    for r in (r1, r2, r3):
        _padding_untouched(r)
    # both exchanges run the same per-rank kernels: identical samples, identical update
    assert_close(float(l3), float(l2), 1e-6, "loss: all-reduce vs factor exchange")
    for k in P.KEYS:
        assert_close(P.N(getattr(r3, k)), P.N(getattr(r2, k)), 1e-5, "all-reduce vs factors: " + k, atol=2e-6)
    # 0.5 million Bernoulli comparisons: the 8-chunk launch and the 64-row launches split K differently, so a handful of
    # near-tie samples (|p - u| < 2e-6) may differ between them; each is checked against the oracle with ITS samples at the ties
    same = all(np.array_equal(a, b) for a, b in zip(dec1, dec2))
    if same:
        for k in P.KEYS:
            assert_close(P.N(getattr(r2, k)), P.N(getattr(r1, k)), 1e-5, "factors vs single process: " + k, atol=2e-6)
    for name, rr, ll, dec in (("single process", r1, l1, dec1), ("8 ranks", r2, l2, dec2)):
        st = O.RBMState.create(W0, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95)
        O.set_tie_break([dec[0], dec[1], None], tol=2e-6)
        o = O.train_epoch(st, X, 0, 1, PhiloxStream(77))
        assert O.TIE_BREAK["ties"] < 300 and O.TIE_BREAK["used"] <= 30, dict(O.TIE_BREAK, queue=None)
        O.set_tie_break(None)
        assert_close(float(ll), o, 1e-5, name + ": loss vs oracle")
        for k in P.KEYS:
            assert_close(P.N(getattr(rr, k)), getattr(st, k), 1e-4, name + " vs oracle: " + k, atol=2e-6)


@pytest.mark.parametrize("V,H,B", [(1041, 132, 128), (4200, 160, 128), (4100, 96, 256), (4100, 200, 256), (4321, 1500, 200),
                                   (10000, 1500, 256), (640, 96, 192), (4500, 332, 180), (4100, 72, 300)])
def test_decode_of_a_multi_chunk_batch_one_block_per_weight_tile(_native, V, H, B):
    """visible_probs / backward (rbm.py:148-151) of more than 64 real-valued rows: wide layers take the LDS-tiled kernel (128 weight
    rows x 64 batch rows per block, activation terms staged once per block); without it one block per 32-row weight tile takes 2 or
    4 batch chunks; without that, one block per (tile, chunk).  Same products, K dealt differently to the waves: equal within fp32
    summation order, and to the fp64 value of sigmoid(h W^T + b)."""
    from imdbn.models import RBM
    g = np.random.Generator(np.random.PCG64(V + B))
    W0 = (g.standard_normal((V, H), dtype=F32) / F32(np.sqrt(H))).astype(F32)
    vb = (g.standard_normal(V, dtype=F32) * F32(0.3)).astype(F32)
    h = g.random((B, H), dtype=F32)
    r = RBM(V, H, 0.1, 1e-4, 0.5).to(DEV)
    _poison_and_fill(r, W0)
    r.vis_bias.data.copy_(torch.from_numpy(vb))
    ht = P.T(h, DEV)
    a = r.backward(ht)
    _native.set_option("no_down_chunks", 1)
    try:
        b = r.backward(ht)
    finally:
        _native.set_option("no_down_chunks", 0)
    _native.set_option("no_down_tiled", 1)                 # (the chunks-per-block form of the per-tile kernel, without the LDS-tiled one)
    try:
        b2 = r.backward(ht)
    finally:
        _native.set_option("no_down_tiled", 0)
    assert_close(P.N(a), P.N(b2), 2e-6, "LDS-tiled vs chunks per block", atol=1e-7)
    ref = 1.0 / (1.0 + np.exp(-(h.astype(np.float64) @ W0.astype(np.float64).T + vb.astype(np.float64))))
    assert a.shape == (B, V) and torch.isfinite(a).all()
    assert_close(P.N(a), P.N(b), 2e-6, "chunks in one block vs one block per chunk", atol=1e-7)
    assert_close(P.N(a), ref.astype(F32), 1e-5, "vs fp64", atol=1e-6)
    _padding_untouched(r)


@pytest.mark.parametrize("live_k", [None, 16])
def test_config5_cross_reconstruct_full_size_against_oracle(live_k):
    """BASELINE configs[4]: TXT->IMG reconstruction, batch 256, 50-step Gibbs + noisy mean-field anneal with z_class_mean,
    decode 500 -> 1500 -> 10000; default (reference-identical inert best-of-K=5) and live best-of-K=16."""
    from torch.utils.data import DataLoader, TensorDataset
    from imdbn import engine as E
    from imdbn.models import iMDBN
    B, K, Dz, steps = 256, 32, 500, 50
    g = np.random.Generator(np.random.PCG64(51))
    X = (g.random((64, 10000), dtype=F32) > 0.9).astype(F32)
    dl = DataLoader(TensorDataset(torch.from_numpy(X), torch.zeros(64, K)), batch_size=64)
    params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95,
              "LEARNING_RATE_DYNAMIC": True, "CD": 1, "JOINT_LEARNING_RATE": 0.04, "CROSS_GIBBS_STEPS": steps}
    if live_k:
        params.update({"CROSS_LIVE_BEST_OF_K": True, "CROSS_BEST_OF_K": live_k})
    m = iMDBN([10000, 1500, Dz], 256, params=params, dataloader=dl, val_loader=dl, device=torch.device(DEV), num_labels=K)
    sizes = [(10000, 1500), (1500, Dz)]
    img_states = []
    for rb, (v, h) in zip(m.image_idbn.layers, sizes):
        W0 = (g.standard_normal((v, h), dtype=F32) / F32(np.sqrt(v))).astype(F32)
        hb = (g.standard_normal(h, dtype=F32) * F32(0.1)).astype(F32)
        vb = (g.standard_normal(v, dtype=F32) * F32(0.1)).astype(F32)
        P.set_params(rb, DEV, W0, hb, vb)
        img_states.append(O.RBMState.create(W0, 0.1, 1e-4, 0.5, hid_bias=hb, vis_bias=vb))
    V = Dz + K
    W0 = (g.standard_normal((V, 256), dtype=F32) / F32(np.sqrt(V))).astype(F32)
    hb = (g.standard_normal(256, dtype=F32) * F32(0.1)).astype(F32)
    vb = (g.standard_normal(V, dtype=F32) * F32(0.1)).astype(F32)
    P.set_params(m.joint_rbm, DEV, W0, hb, vb)
    joint = O.RBMState.create(W0, 0.04, 1e-4, 0.5, softmax_groups=[(Dz, V)], hid_bias=hb, vis_bias=vb)
    zcm = g.random((K, Dz), dtype=F32)
    m.z_class_mean = P.T(zcm, DEV)
    z = g.random((B, Dz), dtype=F32)
    y = np.eye(K, dtype=F32)[g.integers(0, K, B)]
    with E.use_rng(E.PhiloxRng(seed=61)):
        img, p_y = m._cross_reconstruct(P.T(z, DEV), P.T(y, DEV), steps=steps)
    assert img.shape == (B, 10000) and p_y.shape == (B, K)
    if live_k is None:
        want_img, want_py = O.cross_reconstruct(img_states, joint, z, y, steps, PhiloxStream(61), z_class_mean=zcm)
        assert_close(P.N(p_y), want_py, 1e-4, "p(y | image)", atol=2e-6)
        assert_close(P.N(img), want_img, 2e-4, "image from text", atol=5e-6)
    else:
        # live selection (extension): every row's pick has the lowest free energy among its candidates; the label
        # posterior (computed before the selection) still equals the reference path
        _, want_py = O.cross_reconstruct(img_states, joint, z, y, steps, PhiloxStream(61), z_class_mean=zcm, Kbuf=live_k)
        assert_close(P.N(p_y), want_py, 1e-4, "p(y | image)", atol=2e-6)
        assert torch.isfinite(img).all() and float(img.min()) >= 0.0 and float(img.max()) <= 1.0
