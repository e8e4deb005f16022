"""Evaluation side-car (imdbn.utils.probe_utils) on CPU.

The probe is checked against a plain torch restatement of the reference's loop (probe_utils.py:195-263:
``nn.Linear`` + ``torch.optim.AdamW`` + ``F.cross_entropy``, early stopping on the validation loss with a host
``.item()`` per step) -- test infrastructure only, written here from the reference's description of the algorithm."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from imdbn import engine as E
from imdbn.utils import probe_utils as PU
from oracle_engine import OracleEngine


def _autograd_probe(Xtr, ytr, Xva, yva, n_classes, max_steps, lr, weight_decay, patience, min_delta):
    model = nn.Linear(Xtr.shape[1], n_classes)
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=weight_decay)
    Xtr, Xva = torch.tensor(Xtr), torch.tensor(Xva)
    ytr, yva = torch.tensor(ytr, dtype=torch.long), torch.tensor(yva, dtype=torch.long)
    best, state, stale, steps = float("inf"), None, 0, 0
    for _ in range(max_steps):
        steps += 1
        opt.zero_grad()
        F.cross_entropy(model(Xtr), ytr).backward()
        opt.step()
        with torch.no_grad():
            v = F.cross_entropy(model(Xva), yva).item()
        if v < best - min_delta:
            best, stale = v, 0
            state = {k: t.detach().clone() for k, t in model.state_dict().items()}
        else:
            stale += 1
            if stale >= patience:
                break
    if state is not None:
        model.load_state_dict(state)
    with torch.no_grad():
        pred = model(Xva).argmax(1)
    return float((pred == yva).float().mean()), pred.tolist(), model, steps


def _blobs(n, d, k, seed, spread=1.0):
    g = np.random.default_rng(seed)
    centers = g.normal(size=(k, d)).astype(np.float32) * 2.0
    y = g.integers(0, k, n)
    X = (centers[y] + spread * g.normal(size=(n, d))).astype(np.float32)
    return X, y


@pytest.mark.parametrize("spread,wd,patience,max_steps", [(1.0, 0.0, 20, 300), (3.0, 0.01, 5, 400), (2.0, 0.0, 3, 50)])
def test_device_probe_equals_autograd_adamw_loop(spread, wd, patience, max_steps):
    X, y = _blobs(400, 12, 4, seed=3, spread=spread)
    Xtr, ytr, Xva, yva = X[:300], y[:300], X[300:], y[300:]
    torch.manual_seed(11)
    acc_ref, pred_ref, model, steps_ref = _autograd_probe(Xtr, ytr, Xva, yva, 4, max_steps, 1e-2, wd, patience, 0.0)
    torch.manual_seed(11)
    acc, y_true, y_pred = PU.train_linear_classifier(Xtr, ytr, Xva, yva, torch.device("cpu"), 4, max_steps=max_steps,
                                                     lr=1e-2, weight_decay=wd, patience=patience, sync_every=7)
    assert y_true == yva.tolist()
    # same early-stopping point and parameters (closed-form gradient vs autograd: rounding only)
    agree = np.mean(np.array(y_pred) == np.array(pred_ref))
    assert agree >= 0.99 and abs(acc - acc_ref) <= 0.011, (agree, acc, acc_ref, steps_ref)


def test_bins_split_and_confusion():
    v = torch.tensor([0.0, 0.0, 0.0, 0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    lab, edges = PU.make_bin_labels(v, n_bins=5)
    assert edges.numel() == 6 and torch.all(edges[1:] > edges[:-1]) and lab.min() == 0 and lab.max() == 4
    assert torch.equal(lab, torch.bucketize(v, edges[1:-1].contiguous()))
    y = torch.tensor([0] * 10 + [1] * 5 + [2] * 1 + [3] * 2)
    tr, te = PU.stratified_split(y, test_size=0.2, rng_seed=42)
    assert sorted(tr + te) == list(range(18)) and not set(tr) & set(te)
    counts = lambda idx: [int((y[idx] == c).sum()) for c in range(4)]
    assert counts(te) == [2, 1, 1, 1] and counts(tr) == [8, 4, 0, 1]        # singleton class goes to test; >= 1 stays in train
    assert PU.stratified_split(y, 0.2, 42) == (tr, te) and PU.stratified_split(y, 0.2, 43) != (tr, te)
    cm = PU.confusion_matrix(torch.tensor([0, 1, 1, 2, 2, 2]), torch.tensor([0, 1, 0, 2, 2, 1]), 3)
    assert cm.tolist() == [[1, 0, 0], [1, 1, 0], [0, 1, 2]]
    assert PU._format_bin_names(torch.tensor([0.1, 1.70004, 3.25])) == ["0.1-1.7", "1.7-3.25"]


def test_probes_run_on_engine_embeddings(tmp_path, monkeypatch):
    """End to end on a small stack with the oracle test double: embeddings = iDBN.represent / iMDBN.represent of the
    validation loader (device-resident), targets from model.features, one probe per target."""
    from torch.utils.data import DataLoader, TensorDataset
    from imdbn.models import iDBN, iMDBN
    monkeypatch.chdir(tmp_path)
    E.set_engine_for_testing(OracleEngine())
    try:
        g = np.random.default_rng(0)
        K, n = 4, 96
        yi = np.arange(n) % K
        proto = (g.random((K, 50)) > 0.5).astype(np.float32)
        X = np.abs(proto[yi] - (g.random((n, 50)) > 0.95)).astype(np.float32)
        Y = np.eye(K, dtype=np.float32)[yi]
        dl = DataLoader(TensorDataset(torch.from_numpy(X), torch.from_numpy(Y)), batch_size=32, shuffle=False)
        params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95,
                  "LEARNING_RATE_DYNAMIC": True, "CD": 1}
        feats = {"Cumulative Area": torch.from_numpy(X.sum(1)), "Convex Hull": torch.from_numpy(X[:, :25].sum(1)),
                 "Labels": torch.from_numpy(Y), "density": torch.from_numpy(g.random(n).astype(np.float32))}
        mdl = iMDBN([50, 24], 16, params=params, dataloader=dl, val_loader=dl, device=torch.device("cpu"), num_labels=K)
        mdl.features = feats
        mdl.image_idbn.features = feats
        mdl.image_idbn.val_loader = dl
        emb, f = PU.compute_val_embeddings_and_features(mdl.image_idbn)
        assert emb.shape == (n, 24) and set(f) == {"cum_area", "convex_hull", "labels", "density"}
        assert torch.equal(f["labels"], torch.from_numpy(yi).float())
        assert torch.allclose(emb[:32], mdl.image_idbn.represent(torch.from_numpy(X[:32])))
        emb1, _ = PU.compute_val_embeddings_and_features(mdl.image_idbn, upto_layer=1)
        assert emb1.shape == (n, 24)
        torch.manual_seed(0)
        res = PU.log_linear_probe(mdl.image_idbn, epoch=0, n_bins=4, steps=60, save_csv=True, layer_tag="top")
        assert set(res) == {"top/cum_area", "top/convex_hull", "top/labels", "top/density"}
        for r in res.values():
            assert 0.0 <= r["acc"] <= 1.0 and r["confusion"].shape == (4, 4) and int(r["confusion"].sum()) > 0
            assert len(r["bin_names"]) == 4 and (tmp_path / r["csv"]).exists() or __import__("os").path.exists(r["csv"])
        ej, fj = PU.compute_joint_embeddings_and_features(mdl)
        assert ej.shape == (n, 16)
        resj = PU.log_joint_linear_probe(mdl, epoch=0, n_bins=4, steps=40)
        assert set(resj) == {"joint/cum_area", "joint/convex_hull", "joint/labels", "joint/density"}
        bad = dict(feats); bad["Labels"] = torch.zeros(n - 1)
        mdl.features = bad
        with pytest.raises(RuntimeError, match="length mismatch"):
            PU.compute_joint_embeddings_and_features(mdl)
    finally:
        E.set_engine_for_testing(None)


# ---- pinned by the reference: tests/golden/probe_reference_360.npz (make_fixtures.py case_probe ran the reference's own
# ---- imdbn/utils/probe_utils.py on a stub model) ------------------------------------------------------------------------
def _probe_fixture():
    import json, os
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "probe_reference_360.npz"), allow_pickle=False)
    return fx, json.loads(str(fx["meta"]))


class _StubModel:
    """What the fixture's generator handed the reference: a fixed tanh map as `represent`, four batches as the loader."""
    def __init__(self, fx, device, arch_dir):
        X, A = torch.from_numpy(fx["X"]).to(device), torch.from_numpy(fx["A"]).to(device)
        self.device, self.text_flag, self.wandb_run, self.arch_dir = device, False, None, str(arch_dir)
        self.val_loader = [(X[i:i + 100], torch.zeros(len(X[i:i + 100]), 1)) for i in range(0, len(X), 100)]
        self.represent = lambda x, upto_layer=None: torch.tanh(x @ A)
        self.features = {"Cumulative Area": torch.from_numpy(fx["cum_area"]), "Convex Hull": torch.from_numpy(fx["chull"]),
                         "Labels": torch.nn.functional.one_hot(torch.from_numpy(fx["lab"]), 6).float(),
                         "Density": torch.from_numpy(fx["density"])}


def check_probe_against_reference_fixture(device, tmp_path, min_same=1.0):
    fx, meta = _probe_fixture()
    m = _StubModel(fx, device, tmp_path)
    E_, feats = PU.compute_val_embeddings_and_features(m)
    np.testing.assert_allclose(E_.cpu().numpy(), fx["E"], rtol=1e-5, atol=1e-6)
    assert sorted(feats) == ["convex_hull", "cum_area", "density", "labels"]
    for mkey in ("cum_area", "convex_hull", "labels", "density"):
        y, nc, edges, names = PU._prepare_targets(feats, mkey, meta["n_bins"])
        assert nc == meta["n_bins"] and names == meta["bin_names"][mkey]
        assert np.array_equal(y.cpu().numpy(), fx[f"{mkey}_y"])
        np.testing.assert_allclose(edges.cpu().numpy(), fx[f"{mkey}_edges"], rtol=0, atol=1e-6)
        tr, te = PU.stratified_split(y, test_size=0.2, rng_seed=42)
        assert np.array_equal(np.array(tr), fx[f"{mkey}_train_idx"]) and np.array_equal(np.array(te), fx[f"{mkey}_test_idx"])
        torch.manual_seed(meta["seed"])                      # the nn.Linear initialisation comes from torch's generator
        Ef = torch.from_numpy(fx["E"]).to(device)
        acc, yt, yp = PU.train_linear_classifier(Ef[tr], y[tr], Ef[te], y[te], device, nc, max_steps=meta["steps"], lr=1e-2,
                                                 weight_decay=meta["weight_decay_direct"], patience=meta["patience"], min_delta=0.0)
        assert yt == fx[f"{mkey}_y_true"].tolist()
        same = float(np.mean(np.array(yp) == fx[f"{mkey}_y_pred"]))
        assert same >= min_same, (mkey, same)
        assert abs(acc - float(fx[f"{mkey}_acc"])) <= (1.0 - min_same) + 1e-6, (mkey, acc, float(fx[f"{mkey}_acc"]))
    # the orchestrator end to end: the confusion matrices the reference wrote as CSV
    torch.manual_seed(meta["seed"])
    res = PU.log_linear_probe(m, epoch=3, n_bins=meta["n_bins"], steps=meta["steps"], lr=1e-2, patience=meta["patience"],
                              save_csv=True, layer_tag="top")
    for mkey in ("cum_area", "convex_hull", "labels", "density"):
        cm = res[f"top/{mkey}"]["confusion"].numpy()
        want = fx[f"{mkey}_confusion"]
        assert cm.sum() == want.sum() and np.abs(cm - want).sum() <= 2 * round((1.0 - min_same) * want.sum()), (mkey, cm, want)
        import pandas as pd
        df = pd.read_csv(res[f"top/{mkey}"]["csv"], index_col=0)            # same file name and layout as the reference's CSV
        assert [str(c) for c in df.columns] == meta["bin_names"][mkey] and np.array_equal(df.to_numpy(), cm)
        assert res[f"top/{mkey}"]["csv"].endswith(f"probe_top_{mkey}_confusion_epoch3.csv")


def test_probe_side_car_reproduces_the_reference_run(tmp_path):
    """Binning, stratified split, the AdamW probe with early stopping and the confusion matrices equal what the reference's
    own probe_utils.py produced on the same stub model (CPU: same generator, same arithmetic order -> identical predictions)."""
    check_probe_against_reference_fixture(torch.device("cpu"), tmp_path, min_same=1.0)
