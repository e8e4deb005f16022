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


# ---- the joint model under data parallelism: clamped updates, bias initialisation counters, epoch metrics ----
J_SIZES, J_H, J_K, J_B, J_NB, J_EPOCHS = [60, 30], 20, 4, 16, 2, 9      # 9 epochs: crosses the 8-epoch warm-up


def _make_joint(rank=None, world=1):
    """Small iMDBN with fixed parameters; rank r's loader holds rows [r*per, (r+1)*per) of every global batch."""
    from imdbn.models import iMDBN
    from torch.utils.data import DataLoader, TensorDataset
    g = np.random.Generator(np.random.PCG64(11))
    n = J_B * J_NB
    yi = np.arange(n) % J_K
    proto = (g.random((J_K, J_SIZES[0])) > 0.6).astype(np.float32)
    X = np.abs(proto[yi] - (g.random((n, J_SIZES[0])) > 0.9)).astype(np.float32)
    Y = np.eye(J_K, dtype=np.float32)[yi]
    W_img = (g.standard_normal((J_SIZES[0], J_SIZES[1])) / np.sqrt(J_SIZES[0])).astype(np.float32)
    W_j = (g.standard_normal((J_SIZES[1] + J_K, J_H)) / np.sqrt(J_SIZES[1] + J_K)).astype(np.float32)
    per = J_B // world
    if rank is not None:
        rows = np.concatenate([np.arange(b * J_B + rank * per, b * J_B + (rank + 1) * per) for b in range(J_NB)])
        X, Y = X[rows], Y[rows]
    dl = DataLoader(TensorDataset(torch.from_numpy(X), torch.from_numpy(Y)), batch_size=per if rank is not None else J_B, shuffle=False)
    params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95,
              "LEARNING_RATE_DYNAMIC": True, "CD": 1, "JOINT_CD": 1, "CROSS_GIBBS_STEPS": 4, "JOINT_AUX_COND_STEPS": 10}
    mdl = iMDBN(J_SIZES, J_H, params=params, dataloader=dl, val_loader=dl, device=torch.device("cpu"), num_labels=J_K)
    for r, W in ((mdl.image_idbn.layers[0], W_img), (mdl.joint_rbm, W_j)):
        r.W.data = torch.from_numpy(W.copy())
        r.W_m, r.hb_m, r.vb_m = torch.zeros_like(r.W.data), torch.zeros_like(r.hid_bias.data), torch.zeros_like(r.vis_bias.data)
    return mdl


def _joint_state(mdl):
    jr = mdl.joint_rbm
    h = mdl.joint_history
    return dict(W=jr.W.data.numpy(), hb=jr.hid_bias.data.numpy(), vb=jr.vis_bias.data.numpy(), Wm=jr.W_m.numpy(),
                zcm=mdl.z_class_mean.numpy(), n=np.array([r["n"] for r in h]), top1=np.array([r["text_top1"] for r in h]),
                ce=np.array([r["text_ce"] for r in h]), mse=np.array([r["image_mse"] for r in h]),
                cd=np.array([r["cd_loss"] if r["cd_loss"] is not None else -1.0 for r in h]))


def _joint_worker(rank, world, port, out_dir):
    for p in (ROOT, PKG, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from imdbn import engine as E
    from oracle_engine import OracleEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.chdir(out_dir)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    E.set_engine_for_testing(OracleEngine())
    E.dp.enable(mode="allreduce")
    mdl = _make_joint(rank, world)
    E.set_rng(E.PhiloxRng(SEED))
    mdl.train_joint(J_EPOCHS)
    np.savez(os.path.join(out_dir, f"joint_rank{rank}.npz"), **_joint_state(mdl))
    dist.destroy_process_group()


def test_two_rank_train_joint_equals_single_rank(tmp_path, monkeypatch):
    """train_epoch_clamped, init_joint_bias_from_data's counters and the metric sums of train_joint shard
    by rows -- two ranks on half batches end at the single-process model (draws keyed on the global row)."""
    import socket
    import torch.multiprocessing as mp
    from imdbn import engine as E
    from oracle_engine import OracleEngine
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["PYTHONPATH"] = os.pathsep.join([ROOT, PKG, HERE, os.environ.get("PYTHONPATH", "")])
    mp.spawn(_joint_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    monkeypatch.chdir(tmp_path)
    E.set_engine_for_testing(OracleEngine())
    try:
        mdl = _make_joint()
        with E.use_rng(E.PhiloxRng(SEED)):
            mdl.train_joint(J_EPOCHS)
        ref = _joint_state(mdl)
    finally:
        E.set_engine_for_testing(None)
    a, b = np.load(tmp_path / "joint_rank0.npz"), np.load(tmp_path / "joint_rank1.npz")
    for k in ("W", "hb", "vb", "Wm", "zcm", "n", "top1", "ce", "mse", "cd"):
        np.testing.assert_array_equal(a[k], b[k])                    # replicas and their reports stay identical
    from golden_utils import rel_fro
    np.testing.assert_array_equal(a["n"], ref["n"])
    assert rel_fro(a["zcm"], ref["zcm"]) < 1e-6
    for k, tol in (("W", 2e-5), ("hb", 1e-4), ("vb", 1e-4), ("Wm", 1e-4)):
        assert rel_fro(a[k], ref[k]) < tol, k
    assert np.allclose(a["top1"], ref["top1"]) and np.allclose(a["ce"], ref["ce"], rtol=1e-4) and np.allclose(a["mse"], ref["mse"], rtol=1e-4)
    assert np.allclose(a["cd"], ref["cd"], rtol=1e-4)


def _ragged_worker(rank, world, port, out_dir):
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
    E.dp.enable(mode="allreduce")
    r, X = _make(groups=False)
    E.set_rng(E.PhiloxRng(SEED))
    rows = 8 if rank == 0 else 5                          # an unevenly split ragged batch
    msg = ""
    try:
        r.train_epoch(torch.from_numpy(X[0, :rows]), 0, 10, CD=1)
    except RuntimeError as e:
        msg = str(e)
    E.dp.disable()
    row0_after = E.get_rng().row0
    with open(os.path.join(out_dir, f"ragged{rank}.txt"), "w") as f:
        f.write(f"{row0_after}|{msg}")
    dist.destroy_process_group()


def test_uneven_shards_raise_on_every_rank_instead_of_hanging(tmp_path):
    """ranks with different row counts would pick different block sizes / Philox rows and mis-normalise; the first
    time a row count is seen the ranks agree on it, and a mismatch raises on all of them.  disable() restores the row offset."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["PYTHONPATH"] = os.pathsep.join([ROOT, PKG, HERE, os.environ.get("PYTHONPATH", "")])
    mp.spawn(_ragged_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        row0, msg = open(tmp_path / f"ragged{rank}.txt").read().split("|", 1)
        assert "between 5 and 8 rows" in msg, msg
        assert int(row0) == 0


# ---- iMDBN_BiModal under data parallelism (missing item 7; imdbn_bimodal.py:741-829) ----------------
BM_S1, BM_S2, BM_J, BM_B, BM_NB, BM_EPOCHS = [48, 20], [36, 16], [14, 10], 16, 2, 9       # 9 epochs: crosses the 8-epoch warm-up


def _make_bimodal(rank=None, world=1):
    from imdbn.models import iMDBN_BiModal
    from torch.utils.data import DataLoader, TensorDataset
    g = np.random.Generator(np.random.PCG64(23))
    n = BM_B * BM_NB
    yi = np.arange(n) % 4
    X1 = np.abs((g.random((4, BM_S1[0])) > 0.6).astype(np.float32)[yi] - (g.random((n, BM_S1[0])) > 0.9)).astype(np.float32)
    X2 = np.abs((g.random((4, BM_S2[0])) > 0.6).astype(np.float32)[yi] - (g.random((n, BM_S2[0])) > 0.9)).astype(np.float32)
    Ws = [(g.standard_normal((a, b)) / np.sqrt(a)).astype(np.float32)
          for a, b in ((BM_S1[0], BM_S1[1]), (BM_S2[0], BM_S2[1]), (BM_S1[1] + BM_S2[1], BM_J[0]), (BM_J[0], BM_J[1]))]
    per = BM_B // world
    if rank is not None:
        rows = np.concatenate([np.arange(b * BM_B + rank * per, b * BM_B + (rank + 1) * per) for b in range(BM_NB)])
        X1, X2 = X1[rows], X2[rows]
    dl = DataLoader(TensorDataset(torch.from_numpy(X1), torch.from_numpy(X2)), batch_size=per if rank is not None else BM_B, shuffle=False)
    params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95, "LEARNING_RATE_DYNAMIC": True,
              "CD": 1, "JOINT_CD": 1, "CROSS_GIBBS_STEPS": 4, "JOINT_AUX_COND_STEPS": 10, "JOINT_LEARNING_RATE": 0.05}
    mdl = iMDBN_BiModal(BM_S1, BM_S2, BM_J, params=params, dataloader=dl, val_loader=dl, device=torch.device("cpu"))
    for r, W in zip([mdl.mod1_dbn.layers[0], mdl.mod2_dbn.layers[0], *mdl.joint_layers], Ws):
        r.W.data = torch.from_numpy(W.copy())
        r.W_m, r.hb_m, r.vb_m = torch.zeros_like(r.W.data), torch.zeros_like(r.hid_bias.data), torch.zeros_like(r.vis_bias.data)
    return mdl


def _bimodal_state(mdl):
    out = {}
    for i, r in enumerate(mdl.joint_layers):
        out.update({f"W{i}": r.W.data.numpy(), f"hb{i}": r.hid_bias.data.numpy(), f"vb{i}": r.vis_bias.data.numpy(), f"Wm{i}": r.W_m.numpy()})
    h = mdl.joint_history
    out["mse1"] = np.array([r["mod1_mse"] for r in h]); out["mse2"] = np.array([r["mod2_mse"] for r in h])
    out["cd"] = np.concatenate([r["cd_losses"].numpy() for r in h if r["cd_losses"] is not None])
    return out


def _bimodal_worker(rank, world, port, out_dir):
    for p in (ROOT, PKG, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from imdbn import engine as E
    from oracle_engine import OracleEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.chdir(out_dir)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    E.set_engine_for_testing(OracleEngine())
    E.dp.enable(mode="allreduce")
    mdl = _make_bimodal(rank, world)
    E.set_rng(E.PhiloxRng(SEED))
    mdl.train_joint(BM_EPOCHS)
    np.savez(os.path.join(out_dir, f"bimodal_rank{rank}.npz"), **_bimodal_state(mdl))
    dist.destroy_process_group()


def test_two_rank_bimodal_train_joint_equals_single_rank(tmp_path, monkeypatch):
    """iMDBN_BiModal.train_joint on two ranks (half batches each): the clamped CD-3 warm-up, the per-layer CD through the joint
    stack, the bias-initialisation counters and the per-epoch cross-modal MSE sums end at the single-process run."""
    import socket
    import torch.multiprocessing as mp
    from imdbn import engine as E
    from oracle_engine import OracleEngine
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["PYTHONPATH"] = os.pathsep.join([ROOT, PKG, HERE, os.environ.get("PYTHONPATH", "")])
    mp.spawn(_bimodal_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    monkeypatch.chdir(tmp_path)
    E.set_engine_for_testing(OracleEngine())
    try:
        mdl = _make_bimodal()
        with E.use_rng(E.PhiloxRng(SEED)):
            mdl.train_joint(BM_EPOCHS)
        ref = _bimodal_state(mdl)
    finally:
        E.set_engine_for_testing(None)
    a, b = np.load(tmp_path / "bimodal_rank0.npz"), np.load(tmp_path / "bimodal_rank1.npz")
    for k in ref:
        np.testing.assert_array_equal(a[k], b[k])                    # replicas and their reports stay identical
    from golden_utils import rel_fro
    for i in range(2):
        for k, tol in ((f"W{i}", 5e-5), (f"hb{i}", 2e-4), (f"vb{i}", 2e-4), (f"Wm{i}", 2e-4)):
            assert rel_fro(a[k], ref[k]) < tol, (k, rel_fro(a[k], ref[k]))
    assert np.allclose(a["mse1"], ref["mse1"], rtol=1e-4) and np.allclose(a["mse2"], ref["mse2"], rtol=1e-4)
    assert np.allclose(a["cd"], ref["cd"], rtol=1e-4)
