"""Data-parallel path on CPU: world_size 2 over gloo (127.0.0.1), oracle-backed test double as engine.

Checks the host logic of SURVEY.md 8e: every rank computes statistics on its row shard with the same
pre-update parameters, ONE all-reduce of the packed buffer, identical update with 1/global_batch on
every replica; draws keyed on the GLOBAL row so the sharded run equals the unsharded one.
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "multimodal-idbn_amd")

V, H, B, SEED, STEPS = 96, 40, 16, 77, 3


def _make(groups=True):
    from imdbn.models import RBM
    g = np.random.Generator(np.random.PCG64(5))
    W0 = (g.standard_normal((V, H), dtype=np.float32) / np.float32(np.sqrt(V))).astype(np.float32)
    X = (g.random((STEPS, B, V), dtype=np.float32) > 0.6).astype(np.float32)
    r = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, sparsity=True, sparsity_factor=0.1,
            softmax_groups=[(88, 96)] if groups else None).to("cpu")
    r.W.data = torch.from_numpy(W0.copy())
    r.W_m, r.hb_m, r.vb_m = torch.zeros_like(r.W.data), torch.zeros_like(r.hid_bias.data), torch.zeros_like(r.vis_bias.data)
    return r, X


def _worker(rank, world, port, out_dir, mode="allreduce"):
    for p in (ROOT, PKG, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from imdbn import engine as E
    from oracle_engine import OracleEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    E.set_engine_for_testing(OracleEngine())
    E.dp.enable(mode=mode)
    assert E.dp.active() and E.dp.world_size() == world and E.dp.rank() == rank and E.dp.mode() == mode
    r, X = _make(groups=(mode == "allreduce"))
    E.set_rng(E.PhiloxRng(SEED))
    per = B // world
    losses = []
    for s in range(STEPS):
        shard = torch.from_numpy(X[s, rank * per:(rank + 1) * per])
        losses.append(float(r.train_epoch(shard, 7, 10, CD=2)))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), W=r.W.data.numpy(), hb=r.hid_bias.data.numpy(),
             vb=r.vis_bias.data.numpy(), Wm=r.W_m.numpy(), losses=np.array(losses, np.float32))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "factors"])
def test_two_rank_update_equals_single_rank(tmp_path, mode):
    """mode "factors": the all-gather exchange of per-rank blocks (the test double's block is its packed statistics;
    the real factor block is exercised on the GPU by test_factor_exchange_*)."""
    import socket
    import torch.multiprocessing as mp
    from imdbn import engine as E
    from oracle_engine import OracleEngine
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["PYTHONPATH"] = os.pathsep.join([ROOT, PKG, HERE, os.environ.get("PYTHONPATH", "")])
    mp.spawn(_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True)

    # single-rank reference on the full global batch, same seed
    E.set_engine_for_testing(OracleEngine())
    try:
        r, X = _make(groups=(mode == "allreduce"))
        ref_losses = []
        with E.use_rng(E.PhiloxRng(SEED)):
            for s in range(STEPS):
                ref_losses.append(float(r.train_epoch(torch.from_numpy(X[s]), 7, 10, CD=2)))
    finally:
        E.set_engine_for_testing(None)
    a = np.load(tmp_path / "rank0.npz")
    b = np.load(tmp_path / "rank1.npz")
    for k in ("W", "hb", "vb", "Wm"):
        np.testing.assert_array_equal(a[k], b[k])          # replicas stay bit-identical
    from golden_utils import rel_fro
    assert rel_fro(a["W"], r.W.data.numpy()) < 1e-6
    assert rel_fro(a["hb"], r.hid_bias.data.numpy()) < 1e-5
    assert rel_fro(a["vb"], r.vis_bias.data.numpy()) < 1e-5
    assert np.allclose(a["losses"], np.array(ref_losses, np.float32), rtol=1e-5)


def test_packed_layout_matches_header():
    """Packed statistics layout of include/imdbn_engine.h: [dW V*H][dc H][db V][sum P+ H][sq-err 1][pad to 4]."""
    from imdbn.engine import native
    lib = native.lib()
    for v, h in ((10000, 1500), (532, 256), (37, 19)):
        n = v * h + 2 * h + v + 1
        assert lib.imdbn_packed_delta_floats(v, h) == (n + 3) // 4 * 4
