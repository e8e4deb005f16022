"""CPU-only: the product's host logic (RBM/iDBN/iMDBN method bodies, step schedules, draw order)
against the reference fixtures, with the oracle-backed TEST DOUBLE standing in for the HIP engine.
No claim about the kernels is made here -- those are tested on the GPU in test_parity_gpu.py."""
import pytest
import torch

import parity_cases as P
from imdbn import engine as E
from oracle_engine import OracleEngine


@pytest.fixture(autouse=True)
def _double():
    E.set_engine_for_testing(OracleEngine())
    yield
    E.set_engine_for_testing(None)


def test_c1_host_logic():
    P.case_c1("cpu", rel=2e-5)


def test_joint_small_host_logic():
    P.case_joint_small("cpu", rel=5e-5)


def test_idbn_small_host_logic():
    P.case_idbn_small("cpu", rel=5e-5)


def test_imdbn_small_host_logic():
    P.case_imdbn_small("cpu", rel=2e-4)


def test_bimodal_small_host_logic(tmp_path):
    mdl = P.case_bimodal_small("cpu", rel=3e-4)
    # pickle round trip in the reference's format, and a reference-written pickle loads into these classes
    path = str(tmp_path / "bimodal.pkl")
    mdl.save_model(path)
    from imdbn.models import iMDBN_BiModal
    pl = iMDBN_BiModal.load_model(path, device=torch.device("cpu"))
    assert pl["num_joint_layers"] == 2 and pl["metadata"]["model_type"] == "iMDBN_BiModal"
    import os
    ref = iMDBN_BiModal.load_model(os.path.join(os.path.dirname(__file__), "golden", "ref_bimodal_small.pkl"), device=torch.device("cpu"))
    assert [tuple(r.W.shape) for r in ref["joint_layers"]] == [(36, 24), (24, 12)]
    for a, b in zip(ref["joint_layers"], mdl.joint_layers):
        assert torch.allclose(a.W.cpu(), b.W.detach().cpu(), rtol=0, atol=3e-4)


def test_pretrained_finetune_host_logic():
    P.case_pretrained_finetune("cpu", rel=5e-5)


def test_live_best_of_k_host_logic():
    P.case_live_best_of_k("cpu")


def test_energy_utils_host_logic():
    P.case_class_free_energies("cpu")


def test_batches_helper_is_the_dataloader():
    """imdbn.utils.batches slices a sequential TensorDataset loader and defers to the DataLoader otherwise."""
    from torch.utils.data import DataLoader, TensorDataset
    from imdbn.utils import batches
    X = torch.arange(50.).view(25, 2); Y = torch.arange(25)
    for kw in (dict(batch_size=8), dict(batch_size=8, drop_last=True), dict(batch_size=8, shuffle=True), dict(batch_size=5),
               dict(batch_size=4, collate_fn=lambda b: b)):
        dl = DataLoader(TensorDataset(X, Y), **kw)
        torch.manual_seed(0); a = list(dl)
        torch.manual_seed(0); b = list(batches(dl))
        assert len(a) == len(b)
        if "collate_fn" in kw:
            continue
        for p_, q_ in zip(a, b):
            assert all(torch.equal(x, y) for x, y in zip(p_, q_)), kw


def test_product_refuses_cpu_without_engine():
    """No silent CPU fallback: without the test double a CPU tensor must raise."""
    E.set_engine_for_testing(None)
    from imdbn.models import RBM
    r = RBM(8, 4, 0.1, 1e-4, 0.5).to("cpu")
    with pytest.raises(E.EngineError):
        r.forward(torch.zeros(2, 8))


def test_batches_tag_binary_datasets_and_the_tag_survives_the_flatten():
    """imdbn.utils.batches: a sequential DataLoader over an in-memory TensorDataset is asked ONCE whether its tensors are 0/1 and
    every batch carries the answer; rows_on_device (the loops' `.to(device).view(B, -1).float()`) keeps it on the new view."""
    import torch
    from torch.utils.data import DataLoader, TensorDataset
    from imdbn.utils import batches, rows_on_device
    X = (torch.rand(10, 2, 6) > 0.5).float()
    Y = torch.rand(10, 3)
    got = list(batches(DataLoader(TensorDataset(X, Y), batch_size=4)))
    assert [tuple(b[0].shape) for b in got] == [(4, 2, 6), (4, 2, 6), (2, 2, 6)]
    assert all(b[0]._imdbn_binary is True and b[1]._imdbn_binary is False for b in got)
    flat = rows_on_device(got[0][0], torch.device("cpu"))
    assert flat.shape == (4, 12) and flat._imdbn_binary is True
    X[0, 0, 0] = 0.5                                   # in-place change: the dataset is asked again
    got = list(batches(DataLoader(TensorDataset(X, Y), batch_size=4)))
    assert got[0][0]._imdbn_binary is False
    shuffled = DataLoader(TensorDataset(X, Y), batch_size=4, shuffle=True)      # anything else goes through the DataLoader untouched
    assert not hasattr(next(iter(batches(shuffled)))[0], "_imdbn_binary")
    assert not hasattr(rows_on_device(torch.rand(3, 4), torch.device("cpu")), "_imdbn_binary")
