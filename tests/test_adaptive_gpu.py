"""What a batch contains is found out ON THE DEVICE.

The reference loop is ``for img, _ in dataloader: img.to(device)`` (idbn.py:199-203): a fresh, untagged tensor per step.
The engine used to ask such a tensor "are you 0/1?" with a device reduction and a host synchronisation, cache the answer per
object and, after 64 unknown tensors in a row, stop asking and switch kernels -- the kernel path (and the last bits of the
result) depended on the history of the process.  Now nothing on the host inspects a batch:

* the preparation of the next batch decides per 64-column x 64-row item whether the slim form (bit plane + one bf16 plane) or
  the three-term forms describe it (``PrepArgs::adaptive``),
* the streaming K1 reads every 64-column item from the bit plane or from its bf16 terms (``K1S_ADAPTIVE``) -- the same
  fragments multiplied in the same order, so the same numbers whichever way an item is read,
* real-valued operands (layer >= 2 inputs) take the same kernel (``K1S_REAL``, last-arriver split-K, no ``finish`` launch).

Hence: fresh untagged tensors == tagged tensors == the no-prefetch path, BIT FOR BIT, for 0/1 data, real data and batches that
mix both; and all of them agree with the numpy oracle at the north_star tolerance.
"""
import numpy as np
import pytest
import torch

import oracle.rbm_oracle as O
import parity_cases as P
from golden_utils import assert_close
from oracle.draws import PhiloxStream

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


def _batches(kind, n, B, V, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    out = []
    for i in range(n):
        if kind == "binary":
            x = (g.random((B, V), dtype=F32) > 0.8).astype(F32)
        elif kind == "real":
            x = g.random((B, V), dtype=F32)
        else:
            # mixed: 0/1 pixels, one stretch of grey levels (inexact in bf16), one of exactly-bf16 halves, different per batch;
            # the stretches start and end inside 64-column items
            x = (g.random((B, V), dtype=F32) > 0.8).astype(F32)
            a = int(g.integers(0, V - 400))
            x[:, a:a + 150] = g.random((B, 150), dtype=F32)
            b = int(g.integers(0, V - 100))
            x[int(g.integers(0, B)), b:b + 70] = 0.5
        out.append(x)
    return out


def _rbm(V, H, seed):
    from imdbn.models import RBM
    g = np.random.Generator(np.random.PCG64(seed))
    W0 = (g.standard_normal((V, H), dtype=F32) / F32(np.sqrt(V))).astype(F32)
    r = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(DEV)      # the constructor's row-padded layout
    r.W.data.copy_(P.T(W0, DEV))
    return r, W0


def _same(a, b, what):
    for k in P.KEYS:
        ta, tb = getattr(a, k), getattr(b, k)
        ta, tb = (ta.data if hasattr(ta, "data") else ta), (tb.data if hasattr(tb, "data") else tb)
        assert torch.equal(ta, tb), f"{what}: {k} differs ({int((ta != tb).sum())} elements)"


def _run(r, xs, tag, prefetch, seed=31):
    """One update per batch; `tag`: None = a FRESH untagged tensor per step (what `img.to(device)` yields), True / False = tagged."""
    from imdbn import engine as E
    losses = []
    with E.use_rng(E.PhiloxRng(seed=seed)):
        ts = [P.T(x, DEV) for x in xs]
        for i, t in enumerate(ts):
            nxt = ts[i + 1] if (prefetch and i + 1 < len(ts)) else None
            if tag is not None:
                t._imdbn_binary = tag
                if nxt is not None:
                    nxt._imdbn_binary = tag
            losses.append(r.train_epoch(t, 0, 10, CD=1, next_data=nxt))
    return torch.stack([l.reshape(()) for l in losses]).cpu()


@pytest.mark.parametrize("kind,n", [("binary", 100), ("real", 24), ("mixed", 24)])
def test_fresh_untagged_tensors_equal_the_tagged_run_bit_for_bit(kind, n):
    """100 fresh untagged 0/1 tensors (more than the 64 after which the old host check gave up), real-valued batches and batches
    that mix 0/1 stretches with grey levels: the untagged run with the next-batch prefetch == the untagged run without it ==
    the run whose tensors carry the loader's tag, on every parameter bit and every loss."""
    V, H, B = 2600, 500, 64
    xs = _batches(kind, n, B, V, seed=5)
    ra, _ = _rbm(V, H, 1)
    rb, _ = _rbm(V, H, 1)
    rc, _ = _rbm(V, H, 1)
    la = _run(ra, xs, None, prefetch=True)
    lb = _run(rb, xs, None, prefetch=False)
    lc = _run(rc, xs, {"binary": True, "real": False, "mixed": False}[kind], prefetch=True)
    assert torch.isfinite(la).all()
    assert torch.equal(la, lb) and torch.equal(la, lc)
    _same(ra, rb, "prefetched (item-wise forms) vs prepared in the call (all forms)")
    _same(ra, rc, "untagged vs tagged")


@pytest.mark.parametrize("V,H,B", [(1576, 12, 201), (1437, 32, 170), (2116, 28, 172), (1100, 4, 64), (3000, 36, 33)])
@pytest.mark.parametrize("kind", ["binary", "mixed"])
def test_narrow_hidden_layers_untagged_equal_tagged(V, H, B, kind):
    """Hidden layers of one or two 32-column tiles: the positive-phase K1 has fewer blocks than the update kernel has spans, so the
    per-item forms are not used there (tools/stress_parity.py found the internal error the host used to return): fresh untagged
    tensors with the prefetch == without it == tagged, bit for bit."""
    xs = _batches(kind, 4, B, V, seed=V)
    ra, _ = _rbm(V, H, 3)
    rb, _ = _rbm(V, H, 3)
    rc, _ = _rbm(V, H, 3)
    la = _run(ra, xs, None, prefetch=True)
    lb = _run(rb, xs, None, prefetch=False)
    lc = _run(rc, xs, kind == "binary", prefetch=True)
    assert torch.isfinite(la).all()
    assert torch.equal(la, lb) and torch.equal(la, lc)
    _same(ra, rb, "prefetched vs prepared in the call")
    _same(ra, rc, "untagged vs tagged")


@pytest.mark.parametrize("kind", ["binary", "real", "mixed"])
def test_untagged_batches_match_the_oracle(kind):
    """The same three kinds of batches against the numpy oracle (Philox twin) at the north_star tolerance; the seed is the
    first whose smallest Bernoulli margin |p - u| is above fp32 rounding level, so every case compares."""
    from imdbn import engine as E
    V, H, B = 1664, 260, 64
    xs = _batches(kind, 3, B, V, seed=9)
    r, W0 = _rbm(V, H, 2)
    for seed in range(40, 80):
        st = O.RBMState.create(W0, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95)
        ps = PhiloxStream(seed)
        O.reset_margin()
        lo = [O.train_epoch(st, x, 0, 1, ps) for x in xs]
        if O.BERNOULLI_MARGIN["min"] > 3e-6:
            break
    else:
        pytest.fail("no Philox seed with a comfortable Bernoulli margin")
    l = _run(r, xs, None, prefetch=True, seed=seed)
    assert_close(l.numpy(), np.array(lo, F32), 1e-5, "losses")
    for k in P.KEYS:
        assert_close(P.N(getattr(r, k)), getattr(st, k), 1e-4, k, atol=2e-6)


def test_headline_shape_untagged_equals_tagged_and_a_mixed_batch_equals_the_full_forms():
    """The benchmarked object (RBM(10000, 1500), pitch 1536): 6 updates on fresh untagged 0/1 batches == the tagged run; then a
    batch whose items mix 0/1 pixels with grey levels, prefetched item by item, == the same batch prepared in the call."""
    V, H, B = 10000, 1500, 64
    xs = _batches("binary", 6, B, V, seed=3)
    ra, _ = _rbm(V, H, 4)
    rb, _ = _rbm(V, H, 4)
    la = _run(ra, xs, None, prefetch=True)
    lb = _run(rb, xs, True, prefetch=True)
    assert torch.equal(la, lb)
    _same(ra, rb, "untagged vs tagged, headline shape")
    xm = _batches("mixed", 3, B, V, seed=8)
    lc = _run(ra, xm, None, prefetch=True)
    ld = _run(rb, xm, None, prefetch=False)
    assert torch.isfinite(lc).all() and torch.equal(lc, ld)
    _same(ra, rb, "mixed batches: item-wise forms vs all forms, headline shape")


def test_the_host_never_synchronises_on_a_batch():
    """No `.item()` / reduction per fresh tensor: 40 untagged batches are enqueued while the device is still busy with the first
    ones (the enqueue loop returns long before the stream drains), and the engine holds no per-tensor cache."""
    import time
    from imdbn import engine as E
    V, H, B = 10000, 1500, 64
    r, _ = _rbm(V, H, 6)
    xs = [P.T(x, DEV) for x in _batches("binary", 40, B, V, seed=2)]
    eng = E.get_hip_engine()
    assert not hasattr(eng, "_bin") and not hasattr(eng, "_bin_misses")
    with E.use_rng(E.PhiloxRng(seed=1)):
        for i in range(8):
            r.train_epoch(xs[i], 0, 10, CD=1, next_data=xs[i + 1])
        torch.cuda.synchronize()
        ev = torch.cuda.Event()
        t0 = time.perf_counter()
        for i in range(8, 39):
            r.train_epoch(xs[i].clone(), 0, 10, CD=1, next_data=None)
        ev.record()
        t_enq = time.perf_counter() - t0
        busy = not ev.query()                      # the stream is still working when the host is done enqueueing
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
    assert busy or t_enq < 0.6 * t_all, (t_enq, t_all)


@pytest.mark.parametrize("V,H,B", [(1500, 500, 64), (2600, 320, 100), (4100, 132, 33)])
def test_streaming_k1_for_real_operands_equals_the_two_launch_path(V, H, B, _native):
    """Real-valued operands of K1 (layer-2 inputs, `forward` of probabilities) through k1_stream (last-arriver split-K, fused
    epilogue) against the partial GEMM + finish launches (option no_k1s_real): same products, other summation order."""
    from imdbn import engine as E
    eng = _native
    g = np.random.Generator(np.random.PCG64(V))
    r, _ = _rbm(V, H, 7)
    x = P.T(g.random((B, V), dtype=F32), DEV)
    x._imdbn_binary = False
    a = r.forward(x)
    eng.set_option("no_k1s_real", 1)
    try:
        b = r.forward(x)
    finally:
        eng.set_option("no_k1s_real", 0)
    ref = torch.sigmoid(x.double() @ r.W.data.double() + r.hid_bias.data.double()).float()
    assert_close(P.N(a), P.N(ref), 2e-6, "k1_stream (real operand) vs float64")
    assert_close(P.N(b), P.N(ref), 2e-6, "two-launch path vs float64")
    xu = x.clone()                                  # untagged: per-item choice, all items real -> the same numbers
    assert torch.equal(r.forward(xu), a)


@pytest.mark.parametrize("B,steps", [(64, 12), (256, 50), (37, 5)])
def test_chain_pair_equals_the_two_chains_bit_for_bit(B, steps, _native):
    """iMDBN._cross_reconstruct's IMG->TXT and TXT->IMG chains as ONE engine call (imdbn_rbm_chain_pair: both chains in one launch
    of the row-parallel chain kernel) against the two calls of the reference order (imdbn.py:424-449): same draws, same bits."""
    from imdbn import engine as E
    from imdbn.models import RBM
    eng = _native
    g = np.random.Generator(np.random.PCG64(B))
    Dz, K, H = 500, 32, 256
    V = Dz + K
    r = RBM(V, H, 0.04, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, softmax_groups=[(Dz, V)]).to(DEV)
    r.vis_bias.data.copy_(P.T((g.standard_normal(V, dtype=F32) * F32(0.3)), DEV))
    z = P.T(g.random((B, Dz), dtype=F32), DEV)
    y = torch.eye(K, device=DEV)[torch.from_numpy(g.integers(0, K, B)).to(DEV)]
    vk = torch.zeros(B, V, device=DEV); km = torch.zeros_like(vk); vk[:, :Dz] = z; km[:, :Dz] = 1
    vy = torch.zeros(B, V, device=DEV); ky = torch.zeros_like(vy); vy[:, Dz:] = y; ky[:, Dz:] = 1
    mu = P.T(g.random((B, Dz), dtype=F32), DEV)
    gibbs = dict(v_known=vk, known_mask=km, n_steps=steps, sample_h=False, sample_v=False)
    nmf = dict(v_known=vy, known_mask=ky, n_steps=steps, T0=3.0, T1=1.0, sigma0=0.9, hot_frac=0.7, sharpen_last=3, T_cold_plus=0.9)
    r._mu_pull = {"mu_k": mu, "eta0": 0.15}
    with E.use_rng(E.PhiloxRng(seed=17)) as rng:
        a1, b1 = r._chain_pair(gibbs, nmf)
        used = rng.offset
    with E.use_rng(E.PhiloxRng(seed=17)) as rng:
        r._mu_pull = None
        a2 = r.conditional_gibbs(**gibbs)
        r._mu_pull = {"mu_k": mu, "eta0": 0.15}
        b2 = r.noisy_meanfield_annealed(**nmf)
        assert rng.offset == used
    eng.set_option("no_chain_pair", 1)
    try:
        with E.use_rng(E.PhiloxRng(seed=17)):
            a3, b3 = r._chain_pair(gibbs, nmf)
    finally:
        eng.set_option("no_chain_pair", 0)
    assert torch.isfinite(a1).all() and torch.isfinite(b1).all()
    assert torch.equal(a1, a2) and torch.equal(b1, b2) and torch.equal(a1, a3) and torch.equal(b1, b3)


def test_train_joint_metrics_on_a_second_stream_equal_the_inline_pass(tmp_path):
    """iMDBN.train_joint with its per-batch metrics (_cross_reconstruct, imdbn.py:615-639) overlapped on a second stream against a
    snapshot of the joint RBM == the same loop with the metrics in line: weights, class means and every metric, bit for bit
    (the draws are assigned on the host in program order either way)."""
    from torch.utils.data import DataLoader, TensorDataset
    from imdbn import engine as E
    from imdbn.models import iMDBN
    g = np.random.Generator(np.random.PCG64(11))
    K, B, NB = 8, 16, 5
    yi = g.integers(0, K, B * NB)
    proto = (g.random((K, 1300), dtype=F32) > 0.7).astype(F32)
    X = np.abs(proto[yi] - (g.random((B * NB, 1300), dtype=F32) > 0.9).astype(F32)).astype(F32)
    Y = np.eye(K, dtype=F32)[yi]
    dl = DataLoader(TensorDataset(torch.from_numpy(X).to(DEV), torch.from_numpy(Y).to(DEV)), batch_size=B, shuffle=False)
    out = []
    for overlap in (True, False):
        params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95, "LEARNING_RATE_DYNAMIC": True,
                  "CD": 1, "JOINT_LEARNING_RATE": 0.04, "JOINT_CD": 1, "JOINT_AUX_COND_STEPS": 12, "CROSS_GIBBS_STEPS": 9,
                  "JOINT_METRICS_OVERLAP": overlap}
        torch.manual_seed(5)
        m = iMDBN([1300, 200, 60], 32, params=params, dataloader=dl, val_loader=dl, device=torch.device(DEV), num_labels=K)
        with E.use_rng(E.PhiloxRng(seed=8)):
            m.image_idbn.train(1)
            m.train_joint(10)                    # 8 warm-up epochs + 2 main ones
        torch.cuda.synchronize()
        out.append(m)
    a, b = out
    _same(a.joint_rbm, b.joint_rbm, "joint RBM, overlapped vs inline metrics")
    assert torch.equal(a.z_class_mean, b.z_class_mean)
    for ha, hb in zip(a.joint_history, b.joint_history):
        for k in ("n", "text_top1", "text_top3", "text_ce", "image_mse", "cd_loss"):
            assert ha[k] == hb[k], (k, ha[k], hb[k])
    assert np.isfinite([h["text_ce"] for h in a.joint_history]).all() and a.joint_history[-1]["n"] == B * NB


@pytest.mark.parametrize("V,H,B,wd", [(1089, 480, 64, 14), (1600, 480, 232, 0), (1198, 604, 27, 17)])
def test_chains_of_a_wide_layer_do_not_depend_on_what_the_workspace_held(V, H, B, wd, _native):
    """Chains of a layer wider than 1024 run one launch per half step, the h|v ones through k1_stream (real-valued operand).  Its
    split-K arrival counters live in the caller's workspace, which may hold anything (torch.empty): filled with NaN or with zeros
    before the call, the chain must give the same finite numbers (tools/stress_chains.py found all-NaN outputs)."""
    from imdbn import engine as E
    from imdbn.models import RBM
    g = np.random.Generator(np.random.PCG64(V + B))
    Dz = V - wd
    W0 = (g.standard_normal((V, H), dtype=F32) / F32(np.sqrt(V)) * F32(2.0)).astype(F32)
    r = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, softmax_groups=[(Dz, V)] if wd else None)
    P.set_params(r, DEV, W0, (g.standard_normal(H, dtype=F32) * F32(0.2)).astype(F32), (g.standard_normal(V, dtype=F32) * F32(0.2)).astype(F32))
    vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
    vk[:, :Dz] = g.random((B, Dz), dtype=F32); km[:, :Dz] = 1
    outs = []
    for fill in (float("nan"), 0.0, 1.0e30):
        _native._workspace(torch.device(DEV), V, H, B).view(torch.float32).fill_(fill)
        with E.use_rng(E.PhiloxRng(seed=3)):
            a = r.noisy_meanfield_annealed(P.T(vk, DEV), P.T(km, DEV), n_steps=3)
            b = r.conditional_gibbs(P.T(vk, DEV), P.T(km, DEV), n_steps=2, sample_h=True)
        outs.append((a, b))
    for a, b in outs:
        assert torch.isfinite(a).all() and torch.isfinite(b).all()
    for a, b in outs[1:]:
        assert torch.equal(a, outs[0][0]) and torch.equal(b, outs[0][1])
