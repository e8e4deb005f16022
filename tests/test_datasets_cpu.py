"""Stimulus dataset + device loaders (imdbn.datasets) -- CPU run of the same code path."""
import numpy as np
import pytest
import torch

from imdbn import engine as E
from imdbn.datasets import DeviceLoader, UniformDataset, create_dataloaders_uniform
from oracle_engine import OracleEngine


def _archive(tmp_path, n=120, side=10, K=4, with_density=True):
    g = np.random.default_rng(1)
    lab = (np.arange(n) % K) + 1                       # numerosities 1..K
    img = (g.random((n, side, side)) > 0.8).astype(np.float32)
    d = dict(D=img, N_list=lab, cumArea_list=img.reshape(n, -1).sum(1), CH_list=g.random(n).astype(np.float32))
    if with_density:
        d["density"] = g.random(n).astype(np.float32)
    np.savez(tmp_path / "stimuli.npz", **d)
    return img.reshape(n, -1), lab


def test_dataset_contract_and_split(tmp_path):
    img, lab = _archive(tmp_path)
    tr, va, te = create_dataloaders_uniform(path2data=str(tmp_path), data_name="stimuli.npz", batch_size=16, device="cpu")
    base = tr.dataset.dataset
    assert isinstance(base, UniformDataset) and base.classes == [1, 2, 3, 4] and len(base) == 120
    for field in ("labels", "cumArea_list", "CH_list", "density_list"):
        assert len(getattr(base, field)) == 120
    parts = [set(l.dataset.indices) for l in (tr, va, te)]
    assert sum(len(p) for p in parts) == 120 and not (parts[0] & parts[1]) and not (parts[0] & parts[2]) and not (parts[1] & parts[2])
    for p, want in zip(parts, (24, 3, 3)):             # 30 per class: 24 / 3 / 3 -- every class in every part
        assert [sum(1 for i in p if lab[i] == c) for c in (1, 2, 3, 4)] == [want] * 4
    # sequential validation loader: rows in subset order, fp32 pixels, one-hot labels
    xs, ys = zip(*list(va))
    X, Y = torch.cat(xs), torch.cat(ys)
    idx = list(va.dataset.indices)
    assert X.dtype == torch.float32 and np.array_equal(X.numpy(), img[idx]) and Y.shape == (12, 4)
    assert np.array_equal(Y.argmax(1).numpy() + 1, lab[idx])
    x0, y0 = base[idx[0]]
    assert torch.equal(x0, X[0]) and torch.equal(y0, Y[0])
    # shuffled training loader: a permutation of the subset, different per epoch, reproducible from the seed
    e1 = torch.cat([x for x, _ in tr]); e2 = torch.cat([x for x, _ in tr])
    assert len(tr) == 6 and e1.shape == (96, 100) and not torch.equal(e1, e2)
    assert torch.equal(e1.sum(0), e2.sum(0)) and np.allclose(e1.sum(0).numpy(), img[sorted(parts[0])].sum(0))
    tr_b, _, _ = create_dataloaders_uniform(data_path=str(tmp_path), data_name="stimuli.npz", batch_size=16, device="cpu")
    assert torch.equal(torch.cat([x for x, _ in tr_b]), e1)
    # the script's keyword style, unimodal labels, missing density, missing file
    _, va_u, _ = create_dataloaders_uniform(data_path=str(tmp_path), data_name="stimuli.npz", batch_size=8,
                                            multimodal_flag=False, device="cpu", num_workers=3)
    _, yk = next(iter(va_u))
    assert yk.dtype == torch.long and yk.ndim == 1
    with pytest.raises(FileNotFoundError):
        create_dataloaders_uniform(path2data=str(tmp_path), data_name="nope.npz", device="cpu")


def test_rank_sharded_loader_partitions_each_global_batch(tmp_path):
    _archive(tmp_path)
    base = UniformDataset(str(tmp_path), "stimuli.npz")
    full, _, _ = create_dataloaders_uniform(dataset=base, batch_size=16, device="cpu")
    r0, _, _ = create_dataloaders_uniform(dataset=base, batch_size=8, device="cpu", rank=0, world_size=2)
    r1, _, _ = create_dataloaders_uniform(dataset=base, batch_size=8, device="cpu", rank=1, world_size=2)
    assert len(full) == len(r0) == len(r1)
    for (x, y), (x0, y0), (x1, y1) in zip(full, r0, r1):
        assert torch.equal(x, torch.cat([x0, x1])) and torch.equal(y, torch.cat([y0, y1]))


def test_models_read_features_and_train_from_device_loader(tmp_path, monkeypatch):
    """idbn.py:129-144 feature extraction from the Subset contract, training loops fed by DeviceLoader, probes on top."""
    from imdbn.models import iMDBN
    from imdbn.utils import probe_utils as PU
    _archive(tmp_path)
    monkeypatch.chdir(tmp_path)
    E.set_engine_for_testing(OracleEngine())
    try:
        tr, va, _ = create_dataloaders_uniform(path2data=str(tmp_path), data_name="stimuli.npz", batch_size=32, device="cpu")
        params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95,
                  "LEARNING_RATE_DYNAMIC": True, "CD": 1, "CROSS_GIBBS_STEPS": 3, "JOINT_AUX_COND_STEPS": 10}
        mdl = iMDBN([100, 20], 12, params=params, dataloader=tr, val_loader=va, device=torch.device("cpu"), num_labels=4)
        f = mdl.image_idbn.features
        idx = list(va.dataset.indices)
        assert set(f) == {"Cumulative Area", "Convex Hull", "Labels", "Density"} and mdl.features is f
        assert f["Labels"].tolist() == [float(va.dataset.dataset.labels[i]) for i in idx]
        with E.use_rng(E.PhiloxRng(3)):
            mdl.image_idbn.train(1)
            mdl.train_joint(1)
        assert mdl.joint_history[0]["n"] == 96
        res = PU.log_joint_linear_probe(mdl, epoch=0, n_bins=3, steps=30)
        assert set(res) == {"joint/cum_area", "joint/convex_hull", "joint/labels", "joint/density"}
    finally:
        E.set_engine_for_testing(None)
