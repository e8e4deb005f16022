"""hipGraph capture of engine call sequences (imdbn.engine.graph.CapturedSteps): a replayed graph must produce exactly what the
same calls produce issued one by one -- fresh Philox draws on every replay through the device-resident draw counter."""
import numpy as np
import pytest
import torch

import parity_cases as P

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


def _rbm(V, H, seed, groups=None):
    from imdbn.models import RBM
    g = np.random.Generator(np.random.PCG64(seed))
    r = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, softmax_groups=groups)
    P.set_params(r, DEV, (g.standard_normal((V, H), dtype=F32) / F32(np.sqrt(V))).astype(F32),
                 (g.standard_normal(H, dtype=F32) * F32(0.1)).astype(F32), (g.standard_normal(V, dtype=F32) * F32(0.1)).astype(F32))
    return r


def _same(a, b):
    for k in P.KEYS:
        ta, tb = getattr(a, k), getattr(b, k)
        assert torch.equal(ta.data if hasattr(ta, "data") else ta, tb.data if hasattr(tb, "data") else tb), k


@pytest.mark.parametrize("V,H,B,binary", [(784, 256, 32, True), (1500, 500, 64, False), (2048, 512, 40, True)])
def test_captured_train_epoch_replays_equal_eager_calls(V, H, B, binary):
    from imdbn import engine as E
    g = np.random.default_rng(V)
    mk = (lambda: (g.random((B, V), dtype=F32) > 0.7).astype(F32)) if binary else (lambda: g.random((B, V), dtype=F32))
    Xs = [P.T(mk(), DEV) for _ in range(7)]
    # (untagged tensors: what a batch contains is found out on the device, per 64-column piece, with the same numbers whichever
    #  way a piece is read -- nothing about the inputs has to be declared, and nothing depends on what ran before)
    ra, rb = _rbm(V, H, 3), _rbm(V, H, 3)
    with E.use_rng(E.PhiloxRng(seed=12)):
        la = [float(ra.train_epoch(x, 2, 10, CD=1)) for x in Xs]
    x_static = Xs[0].clone()
    with E.use_rng(E.PhiloxRng(seed=12)) as _:
        step = E.CapturedSteps(lambda: rb.train_epoch(x_static, 2, 10, CD=1))
        lb = []
        for i, x in enumerate(Xs):
            x_static.copy_(x)
            if i == 4:                                   # an eager call in between moves the host cursor: the next replay follows it
                lb.append(float(rb.train_epoch(x_static, 2, 10, CD=1)))
                continue
            lb.append(float(step()))
        assert step.graph is not None and step._draws == 3
    assert la == lb
    _same(ra, rb)


def test_captured_layer_loop_and_clamped_update_equal_eager_calls():
    """The per-batch sequence of a small stack (two updates with the fused forward) plus a clamped update of a joint RBM with a
    softmax group (30-step noisy mean-field chain in the row-parallel chain kernel, categorical draws) as ONE graph."""
    from imdbn import engine as E
    B = 32
    g = np.random.default_rng(5)
    Xs = [P.T((g.random((B, 400), dtype=F32) > 0.8).astype(F32), DEV) for _ in range(5)]
    Ys = [P.T(np.eye(8, dtype=F32)[g.integers(0, 8, B)], DEV) for _ in range(5)]

    def build():
        return _rbm(400, 120, 1), _rbm(120, 60, 2), _rbm(68, 40, 3, groups=[(60, 68)])

    def batch(l1, l2, jr, x, y):
        _, h1 = l1.train_epoch(x, 1, 10, CD=1, return_forward=True)
        loss2, z = l2.train_epoch(h1, 1, 10, CD=1, return_forward=True)
        vk = torch.cat([torch.zeros_like(z), y], 1)
        km = torch.cat([torch.zeros_like(z), torch.ones_like(y)], 1)
        lj = jr.train_epoch(torch.cat([z, y], 1), 1, 10, CD=1)
        lc = jr.train_epoch_clamped(vk, km, 1, 10, CD=1, cond_init_steps=30, sample_h=False, sample_v=False,
                                    reclamp_negative=False, aux_lr_mult=0.3, use_noisy_init=True)
        return torch.stack([loss2.reshape(()), lj.reshape(()), lc.reshape(())])

    a = build()
    with E.use_rng(E.PhiloxRng(seed=4)):
        la = [batch(*a, x, y).cpu() for x, y in zip(Xs, Ys)]
    b = build()
    xs, ys = Xs[0].clone(), Ys[0].clone()
    with E.use_rng(E.PhiloxRng(seed=4)):
        step = E.CapturedSteps(lambda: batch(*b, xs, ys))
        lb = []
        for x, y in zip(Xs, Ys):
            xs.copy_(x); ys.copy_(y)
            lb.append(step().clone().cpu())
    for u, v in zip(la, lb):
        assert torch.equal(u, v)
    for ra, rb in zip(a, b):
        _same(ra, rb)
