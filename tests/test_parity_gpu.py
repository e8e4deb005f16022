"""GPU parity tests: the HIP engine (through the C ABI) against the reference fixtures and the oracle.

Tolerance (north_star): <= 1e-4 relative (Frobenius) on weights; the replayed draws are the ones the
reference consumed, so any larger deviation is a kernel bug or a Bernoulli flip at a sub-1e-6 margin
(the assertion message says which).
"""
import os

import numpy as np
import pytest
import torch

import oracle.rbm_oracle as O
import parity_cases as P
from golden_utils import Fixture, assert_close, init_W, rel_fro
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
    eng = E.get_hip_engine()             # raises if the library is missing: no silent fallback
    cu, arch = eng.device_info()
    assert "gfx950" in arch, arch
    yield eng


def test_c1_fixture_gpu():
    P.case_c1(DEV, rel=1e-4)


def test_joint_small_fixture_gpu():
    P.case_joint_small(DEV, rel=1e-4)


def test_idbn_small_fixture_gpu():
    P.case_idbn_small(DEV, rel=1e-4)


def test_bimodal_small_fixture_gpu():
    P.case_bimodal_small(DEV, rel=3e-4)


def test_imdbn_small_fixture_gpu():
    P.case_imdbn_small(DEV, rel=3e-4)


def test_pretrained_finetune_fixture_gpu():
    P.case_pretrained_finetune(DEV, rel=1e-4)


def _mk_state(V, H, seed=0):
    """The oracle state of _mk(V, H, None, seed) alone (no device object)."""
    g = np.random.Generator(np.random.PCG64(seed))
    W0 = (g.standard_normal((V, H), dtype=F32) / F32(np.sqrt(V))).astype(F32)
    hb = (g.standard_normal(H, dtype=F32) * F32(0.1)).astype(F32)
    vb = (g.standard_normal(V, dtype=F32) * F32(0.1)).astype(F32)
    return O.RBMState.create(W0, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, hid_bias=hb, vis_bias=vb)


def _mk(V, H, groups=None, seed=0, **kw):
    from imdbn.models import RBM
    g = np.random.Generator(np.random.PCG64(seed))
    W0 = (g.standard_normal((V, H), dtype=F32) / F32(np.sqrt(V))).astype(F32)
    hb = (g.standard_normal(H, dtype=F32) * F32(0.1)).astype(F32)
    vb = (g.standard_normal(V, dtype=F32) * F32(0.1)).astype(F32)
    r = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, softmax_groups=groups, **kw)
    P.set_params(r, DEV, W0, hb, vb)
    st = O.RBMState.create(W0, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, softmax_groups=groups,
                           hid_bias=hb, vis_bias=vb, **kw)
    return r, st, g


@pytest.mark.parametrize("V,H,B,groups", [
    (64, 64, 64, None),            # exactly one tile
    (37, 19, 1, None),             # odd sizes, batch 1, unaligned rows (scalar weight loads)
    (130, 70, 33, None),           # ragged tiles, batch not a multiple of 32
    (200, 96, 100, [(190, 200)]),  # batch > 64 (two M blocks), softmax group at the edge
    (300, 128, 64, [(10, 20), (290, 300)]),   # two groups, one straddling nothing, one at the end
    (1000, 260, 64, None),         # several split-K chunks
    (384, 256, 130, None),         # three 64-row batch chunks: first / middle / last passes of the streaming update kernel
    (1100, 132, 200, [(1090, 1100)]),   # four chunks on the split-K + bit-packed-samples path
])
def test_philox_cd_step_matches_oracle(V, H, B, groups):
    """PHILOX mode: device draws == oracle/draws.py:PhiloxStream, so the whole update must agree."""
    from imdbn import engine as E
    r, st, g = _mk(V, H, groups, seed=V + H, sparsity=True, sparsity_factor=0.1)
    X = (g.random((B, V), dtype=F32) > 0.6).astype(F32)
    Xr = g.random((B, V), dtype=F32)
    with E.use_rng(E.PhiloxRng(seed=99)):
        l1 = r.train_epoch(P.T(X, DEV), 2, 10, CD=2)
        l2 = r.train_epoch(P.T(Xr, DEV), 7, 10, CD=1)
    ps = PhiloxStream(99)
    o1 = O.train_epoch(st, X, 2, 2, ps)
    o2 = O.train_epoch(st, Xr, 7, 1, ps)
    assert_close(np.array([float(l1), float(l2)], F32), np.array([o1, o2], F32), 1e-5, "losses")
    for k in P.KEYS:
        assert_close(P.N(getattr(r, k)), getattr(st, k), 1e-4, k, atol=2e-6)


@pytest.mark.parametrize("B", [5, 64])
def test_philox_chains_and_clamped_match_oracle(B):
    from imdbn import engine as E
    V, H, Dz = 150, 48, 140
    r, st, g = _mk(V, H, [(Dz, V)], seed=3)
    z = g.random((B, Dz), dtype=F32)
    y = np.eye(V - Dz, dtype=F32)[np.arange(B) % (V - Dz)]
    vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
    vk[:, Dz:] = y; km[:, Dz:] = 1
    mu = g.random((B, Dz), dtype=F32)
    with E.use_rng(E.PhiloxRng(seed=5)):
        a = r.conditional_gibbs(P.T(vk, DEV), P.T(km, DEV), n_steps=7, sample_h=True, sample_v=True)
        r._mu_pull = {"mu_k": P.T(mu, DEV), "eta0": 0.15}
        b = r.noisy_meanfield_annealed(P.T(vk, DEV), P.T(km, DEV), n_steps=15)
        r._mu_pull = None
        c = r.conditional_gibbs_annealed(P.T(vk, DEV), P.T(km, DEV), n_steps=9, sample_h_until=5, sample_v_every=2)
        l = r.train_epoch_clamped(P.T(vk, DEV), P.T(km, DEV), 1, 10, CD=2, cond_init_steps=12, sample_h=True,
                                  sample_v=True, reclamp_negative=True)
    ps = PhiloxStream(5)
    assert_close(P.N(a), O.conditional_gibbs(st, vk, km, ps, n_steps=7, sample_h=True, sample_v=True), 1e-4, "cg")
    st.mu_pull = {"mu_k": mu, "eta0": 0.15}
    assert_close(P.N(b), O.noisy_meanfield_annealed(st, vk, km, ps, n_steps=15), 1e-4, "nmf")
    st.mu_pull = None
    assert_close(P.N(c), O.conditional_gibbs_annealed(st, vk, km, ps, n_steps=9, sample_h_until=5, sample_v_every=2),
                 1e-4, "cga")
    lo = O.train_epoch_clamped(st, vk, km, 1, ps, CD=2, cond_init_steps=12, sample_h=True, sample_v=True,
                               reclamp_negative=True)
    assert_close(float(l), lo, 1e-4, "clamped loss")
    for k in P.KEYS:
        assert_close(P.N(getattr(r, k)), getattr(st, k), 1e-4, k, atol=2e-6)


@pytest.mark.parametrize("V,H,Dz,B", [(150, 48, 140, 5), (532, 256, 500, 64), (532, 256, 500, 100), (700, 300, 700, 17)])
def test_row_parallel_chain_kernel_matches_per_launch_path(V, H, Dz, B, _native):
    """kernels_chain.hpp (one block per 16 batch rows runs the whole chain) against the one-launch-per-half-step
    path on the same Philox draws: mean-field chains agree to fp32 summation-order noise, and the draw accounting
    is identical.  (Sampled chains are compared against the oracle in test_philox_chains_and_clamped_match_oracle.)"""
    from imdbn import engine as E
    g = np.random.default_rng(11)
    groups = [(Dz, V)] if Dz < V else None
    vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
    if groups:
        vk[:, Dz:] = np.eye(V - Dz, dtype=F32)[np.arange(B) % (V - Dz)]; km[:, Dz:] = 1
    else:
        vk[:, :50] = g.random((B, 50), dtype=F32); km[:, :50] = 1
    mu = g.random((B, min(Dz, V)), dtype=F32)
    outs = []
    for no_kernel in (1, 0):
        _native.set_option("no_chain_kernel", no_kernel)
        try:
            r, st, _ = _mk(V, H, groups, seed=3)
            rng = E.PhiloxRng(seed=9)
            with E.use_rng(rng):
                a = r.conditional_gibbs(P.T(vk, DEV), P.T(km, DEV), n_steps=7, sample_h=False, sample_v=False)
                r._mu_pull = {"mu_k": P.T(mu, DEV), "eta0": 0.15}
                b = r.noisy_meanfield_annealed(P.T(vk, DEV), P.T(km, DEV), n_steps=33)
                r._mu_pull = None
                l = r.train_epoch_clamped(P.T(vk, DEV), P.T(km, DEV), 1, 10, CD=1, cond_init_steps=30, sample_h=False,
                                          sample_v=False, reclamp_negative=False)
            outs.append([P.N(a), P.N(b), float(l)] + [P.N(getattr(r, k)) for k in P.KEYS] + [rng.offset])
        finally:
            _native.set_option("no_chain_kernel", 0)
    assert outs[0][-1] == outs[1][-1], "draw accounting differs"
    for i, (x, w) in enumerate(zip(outs[0][:-1], outs[1][:-1])):
        assert_close(np.asarray(w), np.asarray(x), 2e-5, f"output {i}", atol=2e-6)


@pytest.mark.parametrize("V,H,B", [(150, 48, 5), (532, 256, 64), (1500, 500, 33)])
def test_free_energy_matches_oracle(V, H, B):
    """imdbn_rbm_free_energy against F(v) = -v.b - sum softplus(c + vW) (energy_utils.py:19-28); both K1 forms
    (fused short-K and split-K + finish) are covered by the shapes."""
    r, st, g = _mk(V, H, None, seed=4)
    v = g.random((B, V), dtype=F32)
    v[:, ::3] = (v[:, ::3] > 0.5)
    assert_close(P.N(r.free_energy(P.T(v, DEV))), O.free_energy(st, v), 1e-5, "free energy")


def test_class_free_energies_gpu():
    P.case_class_free_energies(DEV)
    P.case_class_free_energies(DEV, V=532, Dz=500, H=256, B=64)        # 2048 stacked rows: several batch chunks


def test_live_best_of_k_gpu():
    en = P.case_live_best_of_k(DEV, K=16)
    assert en.shape[0] == 16


@pytest.mark.parametrize("B", [100, 200])
def test_stats_then_apply_equals_fused_update_for_multi_chunk_batches(B, _native):
    """Batches of several 64-row chunks: the statistics kernel accumulates over the chunks, the fused update applies
    the same linear decomposition; both must agree (different rounding order only) and match the shard sum."""
    from imdbn import engine as E
    V, H = 640, 192
    g = np.random.default_rng(2)
    X = (g.random((B, V), dtype=F32) > 0.7).astype(F32)
    r1, st, _ = _mk(V, H, None, seed=8)
    r2, _, _ = _mk(V, H, None, seed=8)
    with E.use_rng(E.PhiloxRng(seed=31)):
        l1 = r1.train_epoch(P.T(X, DEV), 0, 1, CD=1)
    packed = _native.cd_stats(r2, P.T(X, DEV), 1, E.PhiloxRng(seed=31))
    half = (B // 2 + 7) // 8 * 8
    s0 = _native.cd_stats(r2, P.T(X[:half], DEV), 1, E.PhiloxRng(seed=31, row0=0))
    s1 = _native.cd_stats(r2, P.T(X[half:], DEV), 1, E.PhiloxRng(seed=31, row0=half))
    assert_close(P.N(s0 + s1), P.N(packed), 2e-6, "shard sum of the statistics", atol=2e-5)
    l2 = _native.apply_delta(r2, packed, B, 0.1, 0.5)
    assert_close(float(l2), float(l1), 1e-5, "loss")
    for k in P.KEYS:
        assert_close(P.N(getattr(r2, k)), P.N(getattr(r1, k)), 1e-5, k, atol=2e-6)
    o = O.train_epoch(st, X, 0, 1, PhiloxStream(31))
    assert_close(float(l1), o, 1e-5, "loss vs oracle")
    for k in P.KEYS:
        assert_close(P.N(getattr(r1, k)), getattr(st, k), 1e-4, k, atol=2e-6)


@pytest.mark.parametrize("sample_h,sample_v,reclamp", [(False, False, True), (True, True, False)])
def test_clamped_stats_then_apply_equals_clamped_step_and_shards_add_up(sample_h, sample_v, reclamp, _native):
    """Data-parallel clamped update: clamped_stats + apply_delta (no sparsity term) reproduces
    clamped_step, and two row shards with Philox keyed on the global row sum to the unsharded statistics."""
    from imdbn import engine as E
    V, H, Dz, B = 276, 96, 260, 48
    r1, st, g = _mk(V, H, [(Dz, V)], seed=4, sparsity=True, sparsity_factor=0.1)
    r2, _, _ = _mk(V, H, [(Dz, V)], seed=4, sparsity=True, sparsity_factor=0.1)
    y = np.eye(V - Dz, dtype=F32)[g.integers(0, V - Dz, B)]
    vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
    vk[:, Dz:] = y; km[:, Dz:] = 1
    kw = dict(CD=2, cond_init_steps=12, sample_h=sample_h, sample_v=sample_v, reclamp_negative=reclamp)
    with E.use_rng(E.PhiloxRng(seed=17)):
        l1 = r1.train_epoch_clamped(P.T(vk, DEV), P.T(km, DEV), 2, 10, **kw)
    init = r2._nmf_steps(12, 3.0, 1.0, 0.9, 2, 0.9, 0.0)
    args = (init, None, 2, sample_h, sample_v, reclamp)
    packed = _native.clamped_stats(r2, P.T(vk, DEV), P.T(km, DEV), *args, E.PhiloxRng(seed=17)).clone()
    s0 = _native.clamped_stats(r2, P.T(vk[:24], DEV), P.T(km[:24], DEV), *args, E.PhiloxRng(seed=17, row0=0)).clone()
    s1 = _native.clamped_stats(r2, P.T(vk[24:], DEV), P.T(km[24:], DEV), *args, E.PhiloxRng(seed=17, row0=24)).clone()
    assert_close(P.N(s0 + s1), P.N(packed), 2e-6, "shard sum of the clamped statistics", atol=2e-5)
    lr, mom = r2._lr_mom(2)
    l2 = _native.apply_delta(r2, packed, B, 0.3 * lr, mom, sparsity=False)
    assert_close(float(l2), float(l1), 1e-5, "loss")
    for k in P.KEYS:
        assert_close(P.N(getattr(r2, k)), P.N(getattr(r1, k)), 1e-5, k, atol=2e-6)


@pytest.mark.parametrize("V,H,R,Bl", [(640, 192, 2, 64), (2048, 512, 3, 64), (640, 192, 4, 40)])
def test_factor_exchange_equals_single_process_and_allreduce_updates(V, H, R, Bl, _native):
    """Data-parallel factor exchange, emulated on one device: R ranks of Bl rows each run cd_factors (Philox keyed
    on the global row), the blocks are "gathered" by copies, apply_factors runs the update kernel once per rank
    block.  Must equal (a) the single-process update of the R*Bl-row batch and (b) the all-reduce path."""
    from imdbn import engine as E
    g = np.random.default_rng(4)
    B = R * Bl
    X = (g.random((B, V), dtype=F32) > 0.75).astype(F32)
    X[:, ::7] = g.random((B, len(range(0, V, 7))), dtype=F32)          # some real-valued columns: three-term planes
    r1, st, _ = _mk(V, H, None, seed=6, sparsity=True, sparsity_factor=0.1)
    r2, _, _ = _mk(V, H, None, seed=6, sparsity=True, sparsity_factor=0.1)
    r3, _, _ = _mk(V, H, None, seed=6, sparsity=True, sparsity_factor=0.1)
    lr, mom = r1._lr_mom(0)
    with E.use_rng(E.PhiloxRng(seed=77)):
        l1 = r1.train_epoch(P.T(X, DEV), 0, 1, CD=1)
    assert _native.factor_mode_ok(r2, Bl)
    gathered = _native.gather_buffer(r2, Bl, R)
    for rk in range(R):
        blk = _native.cd_factors(r2, P.T(X[rk * Bl:(rk + 1) * Bl], DEV), 1, E.PhiloxRng(seed=77, row0=rk * Bl))
        gathered[rk].copy_(blk)
    l2 = _native.apply_factors(r2, gathered, Bl, B, lr, mom)
    packed = None
    for rk in range(R):
        s = _native.cd_stats(r3, P.T(X[rk * Bl:(rk + 1) * Bl], DEV), 1, E.PhiloxRng(seed=77, row0=rk * Bl)).clone()
        packed = s if packed is None else packed + s
    l3 = _native.apply_delta(r3, packed, B, lr, mom)
    assert_close(np.array([float(l2), float(l3)], F32), np.array([float(l1)] * 2, F32), 1e-5, "losses")
    for k in P.KEYS:
        assert_close(P.N(getattr(r2, k)), P.N(getattr(r1, k)), 1e-5, "factors vs single: " + k, atol=2e-6)
        assert_close(P.N(getattr(r3, k)), P.N(getattr(r1, k)), 1e-5, "allreduce vs single: " + k, atol=2e-6)
    o = O.train_epoch(st, X, 0, 1, PhiloxStream(77))
    assert_close(float(l2), o, 1e-5, "loss vs oracle")
    for k in P.KEYS:
        assert_close(P.N(getattr(r2, k)), getattr(st, k), 1e-4, "factors vs oracle: " + k, atol=2e-6)


def _random_cases():
    g = np.random.default_rng(20261004)
    cases = []
    for _ in range(8):
        V = int(g.integers(17, 1400)); H = int(g.integers(5, 400)); B = int(g.integers(1, 150))
        if g.random() < 0.5:
            H = H // 4 * 4 + 4                                   # aligned rows: float4 kernels
        groups = None
        if g.random() < 0.4 and V > 40:
            wd = int(g.integers(2, 33))
            groups = [(V - wd, V)]
        cases.append((V, H, B, groups, int(g.integers(1, 3))))
    return cases


@pytest.mark.parametrize("V,H,B,groups,cd", _random_cases())
def test_random_shapes_match_oracle(V, H, B, groups, cd):
    """Randomly drawn layer / batch sizes (fixed seed): aligned and unaligned rows, short and split-K visible
    dimensions, batches of one to three 64-row chunks, optional softmax group; CD-k update + a mean-field chain."""
    from imdbn import engine as E
    r, st, g = _mk(V, H, groups, seed=V * 7 + H)
    X = g.random((B, V), dtype=F32)
    X[:, : V // 2] = (X[:, : V // 2] > 0.5)
    Dz = groups[0][0] if groups else V // 3
    vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
    vk[:, :Dz] = X[:, :Dz]; km[:, :Dz] = 1
    with E.use_rng(E.PhiloxRng(seed=5)):
        l = r.train_epoch(P.T(X, DEV), 3, 10, CD=cd)
        a = r.noisy_meanfield_annealed(P.T(vk, DEV), P.T(km, DEV), n_steps=6)
        f = r.free_energy(P.T(X, DEV))
    ps = PhiloxStream(5)
    lo = O.train_epoch(st, X, 3, cd, ps)
    assert_close(float(l), lo, 1e-5, "loss")
    for k in P.KEYS:
        assert_close(P.N(getattr(r, k)), getattr(st, k), 1e-4, k, atol=2e-6)
    assert_close(P.N(a), O.noisy_meanfield_annealed(st, vk, km, ps, n_steps=6), 1e-4, "noisy mean-field chain", atol=2e-6)
    assert_close(P.N(f), O.free_energy(st, X), 2e-5, "free energy")


def test_products_are_fp32_exact():
    """bf16x3 split: v@W against float64 must be at fp32 rounding level (not bf16 level)."""
    r, st, g = _mk(2000, 300, None, seed=11)
    x = g.random((64, 2000), dtype=F32)
    logits_gpu = None
    p = P.N(r.forward(P.T(x, DEV)))
    ref = 1.0 / (1.0 + np.exp(-((x.astype(np.float64) @ st.W.astype(np.float64)) + st.hid_bias)))
    assert np.abs(p - ref).max() < 5e-7, np.abs(p - ref).max()
    h = g.random((64, 300), dtype=F32)
    lg = P.N(r.backward(P.T(h, DEV), return_logits=True))
    ref = (h.astype(np.float64) @ st.W.T.astype(np.float64)) + st.vis_bias
    assert np.abs(lg - ref).max() < 2e-5 * np.abs(ref).max() + 2e-6


def test_c2_headline_digest_gpu():
    """Full-size 10000<->1500, batch 64: 3 updates against the reference's digests."""
    from imdbn import engine as E
    from imdbn.models import RBM
    fx = Fixture("c2_rbm10000x1500_cd1_digest.npz")
    m = fx.meta
    s = fx.stream()
    Vv, Hh, B, U = m["V"], m["H"], m["B"], m["updates"]
    r = RBM(Vv, Hh, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95)
    P.set_params(r, DEV, init_W(s, Vv, Hh))
    X = (s.uniform((B * U, Vv)) > 0.9).astype(F32)
    with E.use_rng(E.ReplayRng(s)):
        losses = [float(r.train_epoch(P.T(X[B * i:B * i + B], DEV), 0, 10, CD=1)) for i in range(U)]
    assert_close(np.array(losses, F32), fx["losses"], 2e-5, "losses")
    for k in ("W", "W_m"):
        a = P.N(getattr(r, k))
        assert abs(a.astype(np.float64).sum() - fx[k + "_sum"]) <= 1e-4 * abs(fx[k + "_sum"]) + 1e-2
        assert abs((a.astype(np.float64) ** 2).sum() - fx[k + "_sumsq"]) <= 1e-4 * fx[k + "_sumsq"]
        assert_close(a.ravel()[fx[k + "_probe_idx"]], fx[k + "_probe_val"], 1e-4, k + " probes", atol=1e-6)
    for k in ("hid_bias", "vis_bias", "hb_m", "vb_m"):
        assert_close(P.N(getattr(r, k)), fx[k], 2e-4, k, atol=1e-6)


def test_full_size_properties():
    """Size-independent properties at the headline size: determinism, shard-invariance of the
    statistics (sum over 2 row shards == unsharded, PHILOX keyed on the global row), and the
    fused update == stats + apply."""
    from imdbn import engine as E
    from imdbn.models import RBM
    Vv, Hh, B = 10000, 1500, 64
    g = np.random.Generator(np.random.PCG64(1))
    W0 = (g.standard_normal((Vv, Hh), dtype=F32) * F32(0.01)).astype(F32)
    X = (g.random((B, Vv), dtype=F32) > 0.9).astype(F32)
    eng = E.get_hip_engine()

    def fresh():
        r = RBM(Vv, Hh, 0.1, 1e-4, 0.5)
        return P.set_params(r, DEV, W0)

    ra, rb = fresh(), fresh()
    with E.use_rng(E.PhiloxRng(seed=42)):
        la = ra.train_epoch(P.T(X, DEV), 0, 1)
    with E.use_rng(E.PhiloxRng(seed=42)):
        lb = rb.train_epoch(P.T(X, DEV), 0, 1)
    assert torch.equal(ra.W.data, rb.W.data) and torch.equal(ra.vis_bias.data, rb.vis_bias.data) and float(la) == float(lb)

    rc = fresh()
    full = eng.cd_stats(rc, P.T(X, DEV), 1, E.PhiloxRng(seed=42))
    s0 = eng.cd_stats(rc, P.T(X[:32], DEV), 1, E.PhiloxRng(seed=42, row0=0))
    s1 = eng.cd_stats(rc, P.T(X[32:], DEV), 1, E.PhiloxRng(seed=42, row0=32))
    assert torch.equal(rc.W.data, P.T(W0, DEV)), "cd_stats must not touch parameters"
    assert rel_fro(P.N(s0 + s1), P.N(full)) < 1e-5
    l = eng.apply_delta(rc, s0 + s1, B, 0.1, 0.5)
    assert rel_fro(P.N(rc.W.data), P.N(ra.W.data)) < 1e-5
    assert rel_fro(P.N(rc.hid_bias.data), P.N(ra.hid_bias.data)) < 1e-4
    assert abs(float(l) - float(la)) < 1e-5


def test_caller_may_rebind_and_mutate_parameters():
    """the engine must read parameters at every call."""
    from imdbn import engine as E
    r, st, g = _mk(96, 40, None, seed=2)
    x = g.random((8, 96), dtype=F32)
    a = P.N(r.forward(P.T(x, DEV)))
    r.W = torch.nn.Parameter(r.W.data * 2.0, requires_grad=False)        # re-bind (imdbn.py:326)
    r.hid_bias.data[:] = 0.5                                              # in-place (imdbn.py:291)
    st.W *= 2; st.hid_bias[:] = 0.5
    b = P.N(r.forward(P.T(x, DEV)))
    assert_close(b, O.forward(st, x), 1e-5, "forward after rebind")
    assert not np.allclose(a, b)


def test_abi_errors_are_python_exceptions():
    from imdbn import engine as E
    from imdbn.models import RBM
    r = RBM(32, 16, 0.1, 1e-4, 0.5, softmax_groups=[(0, 4), (4, 8), (8, 12), (12, 16), (16, 20)]).to(DEV)
    with pytest.raises(E.EngineError):
        r.visible_probs(torch.zeros(2, 16, device=DEV))
    r2 = RBM(32, 16, 0.1, 1e-4, 0.5).to(DEV)
    with pytest.raises(E.EngineError):
        r2.train_epoch(torch.zeros(4, 32, device=DEV), 0, 1, CD=0)


def test_pickle_roundtrip_and_reference_pickle(tmp_path):
    """Appendix C: reference-written pickles load into engine-backed classes and run on the GPU;
    our pickles carry only plain tensors/attributes."""
    import pickle
    from imdbn.models import iMDBN
    import os
    from golden_utils import GOLDEN
    payload = iMDBN.load_model(os.path.join(GOLDEN, "ref_imdbn_small.pkl"), device=torch.device(DEV))
    jr = payload["joint_rbm"]
    fx = Fixture("imdbn_small_100_40_20_j16.npz")
    assert_close(P.N(jr.W), fx["joint_W"], 1e-7, "pickled joint W")
    out = jr.forward(torch.rand(4, jr.num_visible, device=DEV))           # momentum buffers stayed on CPU: engine re-homes
    assert out.shape == (4, jr.num_hidden)
    l = jr.train_epoch(torch.rand(4, jr.num_visible, device=DEV), 0, 1)
    assert torch.isfinite(l)
    p = tmp_path / "rt.pkl"
    with open(p, "wb") as f:
        pickle.dump({"layers": [jr]}, f)
    with open(p, "rb") as f:
        back = pickle.load(f)["layers"][0]
    # (the live object additionally holds the engine's cached native descriptor: raw addresses, never pickled)
    assert torch.equal(back.W.data, jr.W.data) and set(vars(back)) == set(vars(jr)) - {"_imdbn_desc"}
    assert "_imdbn_desc" in vars(jr) and "_imdbn_desc" not in vars(back)


def test_dp_path_over_rccl_world1_equals_fused_update():
    """stats -> torch.distributed all_reduce (backend nccl = RCCL) -> apply on the GPU, single rank:
    must reproduce the fused update (same draws, same arithmetic up to summation order)."""
    import socket
    import torch.distributed as dist
    from imdbn import engine as E
    r1, st, g = _mk(512, 200, None, seed=21, sparsity=True, sparsity_factor=0.1)
    r2, _, _ = _mk(512, 200, None, seed=21, sparsity=True, sparsity_factor=0.1)
    X = (g.random((64, 512), dtype=F32) > 0.7).astype(F32)
    with E.use_rng(E.PhiloxRng(seed=8)):
        l1 = r1.train_epoch(P.T(X, DEV), 3, 10, CD=1)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        E.dp.enable(force=True)
        with E.use_rng(E.PhiloxRng(seed=8)):
            l2 = r2.train_epoch(P.T(X, DEV), 3, 10, CD=1)
    finally:
        E.dp.disable()
        dist.destroy_process_group()
    for k in P.KEYS:
        assert_close(P.N(getattr(r2, k)), P.N(getattr(r1, k)), 1e-5, "dp " + k, atol=1e-7)
    assert abs(float(l1) - float(l2)) < 1e-6


def test_probe_side_car_on_the_device_matches_the_reference_run(tmp_path):
    """tests/golden/probe_reference_360.npz (the reference's own probe_utils.py on a stub model): binning, split and bin names
    exactly; the device-resident AdamW probe within 3 % of the reference's predictions (fp32 matmul order differs on the GPU)."""
    from test_probe_utils_cpu import check_probe_against_reference_fixture
    check_probe_against_reference_fixture(torch.device(DEV), tmp_path, min_same=0.97)


def test_probe_side_car_stays_on_the_device():
    """Evaluation side-car: validation embeddings come from the engine and stay in HBM, the
    linear probe runs there and agrees with its CPU run on the same inputs."""
    from torch.utils.data import DataLoader, TensorDataset
    from imdbn.models import iDBN
    from imdbn.utils import probe_utils as PU
    g = np.random.default_rng(0)
    K, n = 4, 256
    yi = np.arange(n) % K
    proto = (g.random((K, 200)) > 0.5).astype(F32)
    X = np.abs(proto[yi] - (g.random((n, 200)) > 0.95)).astype(F32)
    dl = DataLoader(TensorDataset(torch.from_numpy(X), torch.from_numpy(np.eye(K, dtype=F32)[yi])), batch_size=64)
    params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95,
              "LEARNING_RATE_DYNAMIC": True, "CD": 1}
    import os, tempfile
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            net = iDBN([200, 64, 32], params, dl, dl, torch.device(DEV))
            net.features = {"Cumulative Area": X.sum(1), "Convex Hull": X[:, :100].sum(1), "Labels": yi}
            emb, feats = PU.compute_val_embeddings_and_features(net)
            assert emb.is_cuda and emb.shape == (n, 32) and all(t.is_cuda for t in feats.values())
            st = O.RBMState.create(P.N(net.layers[0].W.data), 0.1, 1e-4, 0.5, hid_bias=P.N(net.layers[0].hid_bias.data))
            st2 = O.RBMState.create(P.N(net.layers[1].W.data), 0.1, 1e-4, 0.5, hid_bias=P.N(net.layers[1].hid_bias.data))
            assert_close(P.N(emb), O.forward(st2, O.forward(st, X)), 1e-5, "embeddings vs oracle")
            y, edges = PU.make_bin_labels(feats["labels"], n_bins=4)
            tr, te = PU.stratified_split(y, 0.25, 1)
            tr_t, te_t = torch.as_tensor(tr, device=DEV), torch.as_tensor(te, device=DEV)
            torch.manual_seed(5)
            acc, yt, yp = PU.train_linear_classifier(emb[tr_t], y[tr_t], emb[te_t], y[te_t], torch.device(DEV), 4,
                                                     max_steps=120, return_tensors=True)
            assert yp.is_cuda
            torch.manual_seed(5)
            acc_c, _, yp_c = PU.train_linear_classifier(emb[tr_t].cpu(), y[tr_t].cpu(), emb[te_t].cpu(), y[te_t].cpu(),
                                                        torch.device("cpu"), 4, max_steps=120)
            assert abs(float(acc) - acc_c) <= 0.02 and np.mean(np.array(yp.cpu().tolist()) == np.array(yp_c)) >= 0.97
            res = PU.log_linear_probe(net, epoch=0, n_bins=4, steps=60, save_csv=False)
            assert set(res) == {"cum_area", "convex_hull", "labels"}
        finally:
            os.chdir(cwd)


def test_device_loader_feeds_training_from_hbm(tmp_path):
    """imdbn.datasets: the split lives in HBM as uint8, batches are device tensors, the stack trains
    from them and the validation features line up with the side-car's embeddings."""
    from imdbn import engine as E
    from imdbn.datasets import create_dataloaders_uniform
    from imdbn.models import iDBN
    from imdbn.utils import probe_utils as PU
    g = np.random.default_rng(2)
    n, K = 640, 8
    img = (g.random((n, 100, 100)) > 0.9).astype(np.uint8)
    np.savez(tmp_path / "stimuli.npz", D=img, N_list=(np.arange(n) % K) + 1, cumArea_list=img.reshape(n, -1).sum(1),
             CH_list=g.random(n).astype(F32))
    tr, va, te = create_dataloaders_uniform(path2data=str(tmp_path), data_name="stimuli.npz", batch_size=64, multimodal_flag=False)
    x, y = next(iter(tr))
    assert x.is_cuda and x.dtype == torch.float32 and x.shape == (64, 10000) and y.is_cuda
    assert tr._x.dtype == torch.uint8 and tr._x.is_cuda and set(x.unique().tolist()) <= {0.0, 1.0}
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95,
                  "LEARNING_RATE_DYNAMIC": True, "CD": 1}
        net = iDBN([10000, 256, 64], params, tr, va, torch.device(DEV))
        assert set(net.features) == {"Cumulative Area", "Convex Hull", "Labels"}
        w0 = net.layers[0].W.data.clone()
        with E.use_rng(E.PhiloxRng(1)):
            net.train(1)
        assert torch.isfinite(net.layers[0].W.data).all() and not torch.equal(w0, net.layers[0].W.data)
        emb, feats = PU.compute_val_embeddings_and_features(net)
        assert emb.is_cuda and emb.shape == (len(va.dataset), 64) and feats["labels"].numel() == emb.size(0)
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize("V,H,B", [(10000, 1500, 64), (2048, 512, 40), (8192, 1024, 100)])
def test_next_batch_prefetch_is_bit_identical_and_guards_against_stale_data(V, H, B, _native):
    """RBM.train_epoch(next_data=): the operand forms of the following batch are prepared by extra blocks of the
    first negative-phase launch.  Same bits as the plain sequence; a batch modified in place, a different tensor or a
    different shape is prepared again."""
    from imdbn import engine as E
    g = np.random.default_rng(9)
    Xs = [P.T((g.random((B, V), dtype=F32) > 0.8).astype(F32), DEV) for _ in range(5)]
    Xs[3] = P.T(g.random((B, V), dtype=F32), DEV)                     # an inexact batch: three operand terms
    r1, _, _ = _mk(V, H, None, seed=6)
    r2, _, _ = _mk(V, H, None, seed=6)
    with E.use_rng(E.PhiloxRng(seed=4)):
        l1 = [float(r1.train_epoch(x, 0, 1, CD=1)) for x in Xs]
    d = _native._desc(r2, True)
    ok = _native.prefetch_ok(d, B)
    assert ok == ("no_prefetch" not in os.environ.get("IMDBN_OPTS", ""))      # aligned weight rows: the fused K2 carries the prefetch blocks
    with E.use_rng(E.PhiloxRng(seed=4)):
        l2 = []
        for i, x in enumerate(Xs):
            nxt = Xs[i + 1] if i + 1 < len(Xs) else None
            if i == 1 and nxt is not None:
                nxt = nxt.clone()                                     # hint a DIFFERENT tensor than the one passed next
            l2.append(float(r2.train_epoch(x, 0, 1, CD=1, next_data=nxt)))
            if i == 2:
                Xs[3].mul_(1.0)                                       # in-place op: version bump -> prefetched forms are dropped
            assert (len(_native._pf) == 1) == (ok and nxt is not None)
    assert l1 == l2
    for k in P.KEYS:
        assert torch.equal(getattr(r1, k).data if hasattr(getattr(r1, k), "data") else getattr(r1, k),
                           getattr(r2, k).data if hasattr(getattr(r2, k), "data") else getattr(r2, k)), k
    # the hint really is consumed: with the prefetched tensor overwritten through a raw copy (no version bump) the
    # step uses the stale forms -- documents the caller's contract rather than a desirable property
    if ok:
        r3, _, _ = _mk(V, H, None, seed=6)
        r4, _, _ = _mk(V, H, None, seed=6)
        a, b = Xs[0].clone(), Xs[1].clone()
        with E.use_rng(E.PhiloxRng(seed=4)):
            r3.train_epoch(a, 0, 1, CD=1, next_data=b)
            assert len(_native._pf) == 1
            _native.set_option("no_prefetch", 0)                      # clears the engine's prefetch state
            assert len(_native._pf) == 0


@pytest.mark.parametrize("V,H,B,binary", [(10000, 1500, 64, True), (1500, 500, 64, False), (2048, 512, 40, True),
                                          (777, 45, 100, True), (4099, 130, 33, False), (4096, 258, 70, True)])
def test_update_with_fused_forward_equals_the_two_calls(V, H, B, binary, _native):
    """train_epoch(return_forward=True) (imdbn_cd_opts.fwd_out: the layer loop's `train_epoch(v); v = forward(v)` as one
    engine call) against the two separate calls: same weights, same loss, same probabilities, bit for bit -- with and
    without the next-batch hint, for 0/1 batches (bit-plane path) and real-valued ones; forward() against the oracle."""
    from imdbn import engine as E
    g = np.random.default_rng(V + B)
    mk = (lambda: (g.random((B, V), dtype=F32) > 0.8).astype(F32)) if binary else (lambda: g.random((B, V), dtype=F32))
    Xs = [P.T(mk(), DEV) for _ in range(3)]
    r1, st, _ = _mk(V, H, None, seed=5)
    r2, _, _ = _mk(V, H, None, seed=5)
    with E.use_rng(E.PhiloxRng(seed=8)):
        a = []
        for x in Xs:
            l = r1.train_epoch(x, 0, 1, CD=1)
            a.append((float(l), r1.forward(x)))
    with E.use_rng(E.PhiloxRng(seed=8)):
        b = []
        for i, x in enumerate(Xs):
            l, h = r2.train_epoch(x, 0, 1, CD=1, next_data=Xs[i + 1] if i + 1 < len(Xs) and i != 1 else None, return_forward=True)
            b.append((float(l), h))
    for (la, ha), (lb, hb) in zip(a, b):
        assert la == lb and torch.equal(ha, hb)
        assert getattr(hb, "_imdbn_binary", None) is False
    for k in P.KEYS:
        ta, tb = getattr(r1, k), getattr(r2, k)
        assert torch.equal(ta.data if hasattr(ta, "data") else ta, tb.data if hasattr(tb, "data") else tb), k
    # forward() itself against the oracle's sigmoid(v W + c) with the engine's final weights
    W, c = P.N(r2.W), P.N(r2.hid_bias)
    want = 1.0 / (1.0 + np.exp(-(P.N(Xs[-1]).astype(np.float64) @ W.astype(np.float64) + c.astype(np.float64))))
    assert_close(P.N(b[-1][1]), want.astype(F32), 1e-5, "forward", atol=1e-6)


@pytest.mark.parametrize("binary", [False, True])
def test_factor_wire_form_round_trips_and_poisons_on_a_false_promise(binary, _native):
    """Wire form of the factor block (bit-packed visible planes): pack -> (all-gather) -> unpack gives apply_factors
    the same bits as the full blocks; a data plane declared binary that is not turns the update into NaN."""
    from imdbn import engine as E
    V, H, B, R = 2048, 512, 40, 3
    g = np.random.default_rng(12)
    Xs = [P.T((g.random((B, V), dtype=F32) > 0.7).astype(F32), DEV) for _ in range(R)]
    r0, _, _ = _mk(V, H, None, seed=2)
    blocks = [_native.cd_factors(r0, Xs[k], 1, E.PhiloxRng(seed=9, row0=k * B)).clone() for k in range(R)]
    wires = torch.stack([_native.pack_factors(r0, b, B, binary).clone() for b in blocks])
    assert wires.size(1) == _native.compact_bytes(V, H, B, binary) < blocks[0].numel() * (0.5 if binary else 0.9)
    ra, _, _ = _mk(V, H, None, seed=2)
    rb, _, _ = _mk(V, H, None, seed=2)
    la = _native.apply_factors(ra, torch.stack(blocks), B, B * R, 0.1, 0.5)
    lb = _native.apply_factors(rb, _native.unpack_factors(rb, wires, B, binary), B, B * R, 0.1, 0.5)
    rw, _, _ = _mk(V, H, None, seed=2)             # head read from the wire blocks in place, planes unpacked
    w2 = wires.clone()
    lw = _native.apply_factors_wire(rw, w2, _native.unpack_factors(rw, w2, B, binary, planes_only=True), B, B * R, 0.1, 0.5)
    assert float(la) == float(lb) == float(lw)
    for k in P.KEYS:
        ta, tb, tw = getattr(ra, k), getattr(rb, k), getattr(rw, k)
        d = lambda t: t.data if hasattr(t, "data") else t
        assert torch.equal(d(ta), d(tb)) and torch.equal(d(ta), d(tw)), k
    if binary:
        bad = _native.cd_factors(r0, P.T(g.random((B, V), dtype=F32), DEV), 1, E.PhiloxRng(seed=9)).clone()
        w = torch.stack([_native.pack_factors(r0, bad, B, True).clone()])
        rc, _, _ = _mk(V, H, None, seed=2)
        _native.apply_factors(rc, _native.unpack_factors(rc, w, B, True), B, B, 0.1, 0.5)
        assert torch.isnan(rc.hid_bias.data).any()
        rd, _, _ = _mk(V, H, None, seed=2)
        _native.apply_factors_wire(rd, w, _native.unpack_factors(rd, w, B, True, planes_only=True), B, B, 0.1, 0.5)
        assert torch.isnan(rd.hid_bias.data).any()


@pytest.mark.parametrize("V,H,B,binary", [(2048, 512, 40, True), (2048, 512, 40, False), (10000, 1500, 64, True)])
def test_fused_dp_halves_equal_the_separate_calls(V, H, B, binary, _native):
    """imdbn_rbm_cd_factors_wire (CD pass + pack, with the next-batch hint) and imdbn_rbm_apply_wire (unpack + update) against
    cd_factors / pack_factors / unpack_factors / apply_factors_wire: two emulated ranks, four steps, the hint naming the batch
    of the following CD pass (so the data-side factors are packed from a prefetch slot): same wire blocks, same weights."""
    from imdbn import engine as E
    R, T = 2, 4
    g = np.random.default_rng(V + B)
    mk = lambda: P.T((g.random((B, V), dtype=F32) > 0.7).astype(F32), DEV)
    Xs = [[mk() for _ in range(R)] for _ in range(T)]
    if not binary:
        Xs[2][1] = P.T(g.random((B, V), dtype=F32), DEV)                # an inexact batch: three-term planes on the wire
    seq = [Xs[t][k] for t in range(T) for k in range(R)]
    ra, _, _ = _mk(V, H, None, seed=2)
    rb, _, _ = _mk(V, H, None, seed=2)
    rng_a = [E.PhiloxRng(seed=9, row0=k * B) for k in range(R)]
    rng_b = [E.PhiloxRng(seed=9, row0=k * B) for k in range(R)]
    used_slot = 0
    for t in range(T):
        wa = []
        for k in range(R):
            blk = _native.cd_factors(ra, Xs[t][k], 1, rng_a[k])
            wa.append(_native.pack_factors(ra, blk, B, binary).clone())
        wa = torch.stack(wa)
        la = _native.apply_factors_wire(ra, wa, _native.unpack_factors(ra, wa, B, binary, planes_only=True), B, B * R, 0.1, 0.5)
        wb = []
        for k in range(R):
            i = t * R + k
            nxt = seq[i + 1] if i + 1 < len(seq) else None
            before = len(_native._pf)
            wb.append(_native.cd_factors_wire(rb, Xs[t][k], 1, rng_b[k], binary, next_data=nxt).clone())
            used_slot += before
        wb = torch.stack(wb)
        assert torch.equal(wa[:, : wa.size(1) - 256], wb[:, : wb.size(1) - 256]), f"wire blocks differ at step {t}"   # (trailer = pack epoch)
        lb = _native.apply_wire(rb, wb, B, B * R, binary, 0.1, 0.5)
        assert float(la) == float(lb)
        for k in P.KEYS:
            ta, tb = getattr(ra, k), getattr(rb, k)
            assert torch.equal(ta.data if hasattr(ta, "data") else ta, tb.data if hasattr(tb, "data") else tb), (k, t)
    if _native.prefetch_ok(_native._desc(rb, True), B):
        assert used_slot >= T * R - 1              # the hints were taken: the passes after the first read a prefetch slot
    _native._pf.clear()


@pytest.mark.parametrize("V,H,B,rows,kernel", [
    (545, 380, 127, 24, "old"), (20, 468, 95, 24, "old"), (982, 24, 52, 28, "old"), (545, 64, 33, 20, "old"),
    (5511, 64, 33, 0, "old"), (6301, 36, 20, 0, "old"), (4500, 40, 70, 0, "old"),
    (545, 380, 127, 40, "k2s"), (2000, 468, 95, 8, "k2s"), (982, 24, 52, 24, "k2s"), (5511, 64, 33, 48, "k2s"),
    (6301, 36, 20, 0, "k2s"), (1201, 132, 70, 16, "k2s"), (4500, 40, 130, 32, "k2s")])
def test_fused_k2_row_tiles_that_do_not_end_on_16_columns(V, H, B, rows, kernel, _native):
    """Tile heights of the fused K2 kernels (rows = 0: the automatic choice; else forced), old = gemm_down_fused (20 / 24 /
    28-row tiles for V in (4096, 7168]), k2s = k2_stream (8 .. 48 rows, one / two / three 16-row MFMA tiles): the tiles must
    cover the operand forms' padding columns [V, Vpad) -- two consecutive updates and a chain (the second propagation reads
    what the first left) against the oracle.  Regression for a bug found by tools/stress_parity.py (stale padding columns
    -> NaN).  EVERY case compares: the Philox seed is the first one for which the oracle's smallest Bernoulli margin |p - u|
    is above rounding level (1e-6; fp32 summation-order noise in p is ~1e-7), so no case silently degrades to a finiteness check."""
    from imdbn import engine as E
    if kernel == "old":
        _native.set_option("no_k2s", 1); _native.set_option("down_rows", rows)
    else:
        _native.set_option("k2s_rows", rows)
    try:
        g0 = np.random.Generator(np.random.PCG64(V + H))
        _, st0, g = _mk(V, H, None, seed=V + H)
        Xs = [(g.random((B, V), dtype=F32) > 0.6).astype(F32) for _ in range(2)]
        vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
        vk[:, : V // 2] = Xs[0][:, : V // 2]; km[:, : V // 2] = 1
        seed = want = None
        for cand in range(3, 100):
            st = _mk_state(V, H, seed=V + H)
            O.reset_margin()
            s = PhiloxStream(cand)
            o0 = O.train_epoch(st, Xs[0], 1, 2, s); o1 = O.train_epoch(st, Xs[1], 1, 1, s)
            oo = O.conditional_gibbs(st, vk, km, s, n_steps=2)
            if O.BERNOULLI_MARGIN["min"] > 1e-6:
                seed, want = cand, (o0, o1, oo, st)
                break
        assert seed is not None, "no Philox seed with a comfortable Bernoulli margin among 97 candidates"
        o0, o1, oo, st = want
        r, _, _ = _mk(V, H, None, seed=V + H)
        # poison the workspace first: stale contents must not matter
        ws = _native._workspace(torch.device(DEV), V, H, B)
        ws.view(torch.float32).fill_(float("nan"))
        with E.use_rng(E.PhiloxRng(seed=seed)):
            l0 = float(r.train_epoch(P.T(Xs[0], DEV), 1, 10, CD=2))
            l1 = float(r.train_epoch(P.T(Xs[1], DEV), 1, 10, CD=1))
            out = P.N(r.conditional_gibbs(P.T(vk, DEV), P.T(km, DEV), n_steps=2))
        assert np.isfinite([l0, l1]).all() and np.isfinite(out).all()
        assert_close(np.array([l0, l1], F32), np.array([o0, o1], F32), 1e-5, "losses")
        for k in P.KEYS:
            assert_close(P.N(getattr(r, k)), getattr(st, k), 1e-4, k, atol=2e-6)
        assert_close(out, oo, 1e-4, "chain", atol=2e-6)
    finally:
        for k in ("no_k2s", "down_rows", "k2s_rows"):
            _native.set_option(k, 0)


def test_idbn_train_lookahead_with_a_ragged_last_batch_equals_plain_loop(tmp_path):
    """iDBN.train hands the first layer its next batch (operand forms prefetched); a smaller last batch, a second
    layer without a hint and the epoch boundary must not change any bit."""
    from torch.utils.data import DataLoader, TensorDataset
    from imdbn import engine as E
    from imdbn.models import iDBN
    g = np.random.default_rng(4)
    X = torch.from_numpy((g.random((164, 4096)) > 0.8).astype(F32)).to(DEV)          # batches of 64, 64, 36
    dl = DataLoader(TensorDataset(X, torch.zeros(len(X), 1, device=DEV)), batch_size=64)
    params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95,
              "LEARNING_RATE_DYNAMIC": True, "CD": 1}
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        torch.manual_seed(0); a = iDBN([4096, 256, 64], params, dl, dl, torch.device(DEV))
        torch.manual_seed(0); b = iDBN([4096, 256, 64], params, dl, dl, torch.device(DEV))
        for la, lb in zip(a.layers, b.layers):
            assert torch.equal(la.W.data, lb.W.data)
        with E.use_rng(E.PhiloxRng(seed=6)):
            a.train(2)
        with E.use_rng(E.PhiloxRng(seed=6)):                       # the same loop without any hint
            for epoch in range(2):
                for s in range(0, 164, 64):
                    v = X[s:s + 64]                 # untagged: the device finds out what the batch contains; same numbers either way
                    for rbm in b.layers:
                        rbm.train_epoch(v, epoch, 2, CD=1)
                        v = rbm.forward(v)
        for la, lb in zip(a.layers, b.layers):
            for k in P.KEYS:
                ta, tb = getattr(la, k), getattr(lb, k)
                assert torch.equal(ta.data if hasattr(ta, "data") else ta, tb.data if hasattr(tb, "data") else tb), k
    finally:
        os.chdir(cwd)


def test_options_handle_overrides_the_defaults_for_the_bound_thread_only(_native):
    """imdbn_options / imdbn_use_options: the knobs of a bound handle act like the same imdbn_set_option settings (bit for
    bit), and unbinding restores the process defaults untouched."""
    from imdbn import engine as E
    V, H, B = 2048, 512, 64
    g = np.random.default_rng(3)
    Xs = [P.T((g.random((B, V), dtype=F32) > 0.8).astype(F32), DEV) for _ in range(2)]

    def run():
        r, _, _ = _mk(V, H, None, seed=4)
        with E.use_rng(E.PhiloxRng(seed=3)):
            ls = [float(r.train_epoch(x, 0, 1, CD=1)) for x in Xs]
        return ls, [P.N(getattr(r, k)) for k in P.KEYS]

    base = run()
    h = _native.options_create(no_k1s=1, no_k2s=1)
    try:
        _native.use_options(h)
        bound = run()
        _native.use_options(None)
        again = run()
    finally:
        _native.use_options(None)
        _native.options_destroy(h)
    _native.set_option("no_k1s", 1); _native.set_option("no_k2s", 1)
    try:
        glob = run()
    finally:
        _native.set_option("no_k1s", 0); _native.set_option("no_k2s", 0)
    assert bound[0] == glob[0] and all(np.array_equal(a, b) for a, b in zip(bound[1], glob[1]))
    assert again[0] == base[0] and all(np.array_equal(a, b) for a, b in zip(again[1], base[1]))
    from golden_utils import rel_fro
    assert rel_fro(bound[1][0], base[1][0]) < 1e-3          # a different kernel path: the same update up to summation order / a near-tie
    assert any(not np.array_equal(a, b) for a, b in zip(bound[1], base[1])), "the bound knobs had no effect"


@pytest.mark.parametrize("V,H,B", [(300, 128, 33), (2048, 512, 64), (777, 45, 100)])
def test_assoc_update_alone_matches_oracle(V, H, B, _native):
    """imdbn_rbm_assoc_update (K3 on its own): the weight / bias update of rbm.py:209-224 from caller tensors."""
    r, st, g = _mk(V, H, None, seed=V + 3 * H, sparsity=True, sparsity_factor=0.1)
    vpos = (g.random((B, V), dtype=F32) > 0.7).astype(F32)
    vneg = g.random((B, V), dtype=F32)                                    # real-valued: three-term planes
    hpos, hneg = g.random((B, H), dtype=F32), g.random((B, H), dtype=F32)
    lr, mom = r._lr_mom(7)
    _native.assoc_update(r, P.T(vpos, DEV), P.T(hpos, DEV), P.T(vneg, DEV), P.T(hneg, DEV), lr, mom)
    s = dict(pos_assoc=(vpos.T @ hpos).astype(F32), neg_assoc=(vneg.T @ hneg).astype(F32), pos_h_sum=hpos.sum(0, dtype=F32),
             neg_h_sum=hneg.sum(0, dtype=F32), data_sum=vpos.sum(0, dtype=F32), v_sum=vneg.sum(0, dtype=F32))
    O.apply_cd_update(st, s, lr, mom, B, True)
    for k in P.KEYS:
        assert_close(P.N(getattr(r, k)), getattr(st, k), 2e-5, k, atol=2e-6)


def test_rccl_through_the_c_abi_world_of_one(_native):
    """imdbn_comm_* / imdbn_allreduce_sum_f32 / imdbn_allgather_bytes: a communicator of one rank over librccl (the multi-rank
    flow is the caller's: carry the 128-byte id from rank 0 to the others, then the same calls)."""
    uid = _native.comm_unique_id()
    assert len(uid) == 128
    comm = _native.comm_init(1, 0, uid)
    try:
        t = torch.arange(1000, device=DEV, dtype=torch.float32)
        _native.comm_allreduce_sum(comm, t)
        blk = torch.arange(4096, device=DEV, dtype=torch.uint8)
        out = torch.zeros(4096, device=DEV, dtype=torch.uint8)
        _native.comm_allgather(comm, blk, out)
        torch.cuda.synchronize()
        assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32)) and torch.equal(out, blk)
    finally:
        _native.comm_destroy(comm)


def test_bench_multi_gpu_form_starts_its_own_ranks():
    """`python bench.py --gpus 2` typed as is (no torchrun): the parent starts the ranks as child processes before touching the
    GPU and relays rank 0's JSON line.  Rehearsed with two ranks sharing this one GPU over gloo (RCCL needs one GPU per rank);
    both exchanges are timed in the run and the replicas must stay bit-identical."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6", "--warmup", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["replicas_identical"] is True and d["config"]["dp_exchange"].startswith("factors")
    o = d["dp_other_exchange"]
    assert o["dp_exchange"].startswith("allreduce") and o["replicas_identical"] is True and o["value"] > 0
