"""Evaluation side-car kept on the device (reference ``imdbn/utils/probe_utils.py``).

Same entry points as the reference module -- ``compute_val_embeddings_and_features`` (:21-80),
``compute_joint_embeddings_and_features`` (:84-135), ``make_bin_labels`` (:141-156), ``stratified_split`` (:170-189),
``train_linear_classifier`` (:195-263), ``log_linear_probe`` (:344-433), ``log_joint_linear_probe`` (:435-510) -- with
the data flow re-laid for the accelerator:

* the validation embeddings are ``represent`` calls on the engine (one K1 launch chain per batch), they are
  concatenated ON THE DEVICE and never visit the host (the reference moves every batch to the CPU and the probe moves
  them back);
* the linear probe is a closed-form soft-max regression step (no autograd graph): logits, gradient, AdamW moments
  and the early-stopping bookkeeping (best loss, best parameters, patience counter) are device tensors, so a whole
  probe runs without a host synchronisation except one flag read every ``sync_every`` steps; steps issued after the
  stopping point are masked out, which makes the result identical to stopping exactly there;
* confusion matrices are one ``bincount``.

Logging to Weights & Biases / matplotlib is observability and stays out of scope: a ``wandb_run`` object, when the model
has one, only receives plain scalars and tables through ``.log``.
"""
from __future__ import annotations

import os
import random
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .batches import batches

_FEATURE_KEYS = (("cum_area", ("Cumulative Area", "cum_area")),
                 ("convex_hull", ("Convex Hull", "convex_hull", "convexhull")),
                 ("labels", ("Labels", "labels")),
                 ("density", ("Density", "density")))


def _norm(k: str) -> str:
    return k.lower().replace(" ", "").replace("_", "")


def _features(model, n: int, device) -> Dict[str, torch.Tensor]:
    """model.features -> {"cum_area", "convex_hull", "labels"[, "density"]} as 1-D float tensors of length n
    (key lookup ignores case / blanks / underscores, one-hot labels become indices; probe_utils.py:33-78)."""
    src = getattr(model, "features", None)
    if src is None:
        raise RuntimeError("model.features is required")
    by_norm = {_norm(k): k for k in src.keys()}
    out = {}
    for name, aliases in _FEATURE_KEYS:
        key = next((by_norm[_norm(a)] for a in aliases if _norm(a) in by_norm), None)
        if key is None or src[key] is None:
            continue
        t = torch.as_tensor(src[key])
        if t.ndim == 2:
            t = torch.argmax(t, dim=1)
        t = t.reshape(-1).to(device=device, dtype=torch.float32)
        if t.numel() != n:
            raise RuntimeError(f"Feature '{name}' length mismatch: {t.numel()} vs embeddings {n}.")
        out[name] = t
    return out


@torch.no_grad()
def compute_val_embeddings_and_features(model, upto_layer: Optional[int] = None) -> Tuple[torch.Tensor, dict]:
    """Embeddings of the whole validation loader, [N, D] on ``model.device``, and the feature targets."""
    assert model.val_loader is not None, "val_loader is None."
    text = getattr(model, "text_flag", False)
    embeds = []
    for data, labels in batches(model.val_loader):
        x = (labels if text else data).to(model.device)
        x = x.view(x.size(0), -1).float()
        embeds.append(model.represent(x) if upto_layer is None else model.represent(x, upto_layer=upto_layer))
    E = torch.cat(embeds, dim=0)
    return E, _features(model, E.size(0), E.device)


@torch.no_grad()
def compute_joint_embeddings_and_features(model) -> Tuple[torch.Tensor, dict]:
    """Joint-layer embeddings ``model.represent((img, labels))`` of the validation loader."""
    assert model.val_loader is not None, "val_loader is None."
    embeds = [model.represent((img.to(model.device), lab.to(model.device))) for img, lab in batches(model.val_loader)]
    if not embeds:
        return torch.empty(0), {}
    E = torch.cat(embeds, dim=0)
    return E, _features(model, E.size(0), E.device)


def make_bin_labels(values: torch.Tensor, n_bins: int = 5):
    """Quantile binning: labels in 0..n_bins-1 and the n_bins+1 edges; equal edges are separated by 1e-6."""
    values = values.float()
    qs = torch.linspace(0, 1, steps=n_bins + 1, device=values.device)
    edges = torch.quantile(values, qs, interpolation="linear")
    e = edges.tolist()                                # n_bins+1 scalars: the only host visit
    for k in range(1, len(e)):
        if e[k] <= e[k - 1]:
            e[k] = e[k - 1] + 1e-6
    edges = torch.tensor(e, dtype=edges.dtype, device=values.device)
    return torch.bucketize(values, edges[1:-1].contiguous(), right=False), edges


def _format_bin_names(edges: torch.Tensor, precision: int = 4) -> List[str]:
    e = edges.detach().cpu().numpy().astype(float)

    def fmt(v):
        return f"{v:.{precision}f}".rstrip("0").rstrip(".")
    return [f"{fmt(e[i])}-{fmt(e[i + 1])}" for i in range(len(e) - 1)]


def stratified_split(labels: torch.Tensor, test_size: float = 0.2, rng_seed: int = 42):
    """Per-class shuffled split over ALL samples (python ``random.Random(rng_seed)``, classes in ascending order, at
    least one sample of every class with > 1 members stays in train); returns (train_idx, test_idx) index lists."""
    rng = random.Random(rng_seed)
    lab = labels.detach().cpu().numpy()
    train_idx: List[int] = []
    test_idx: List[int] = []
    for c in np.unique(lab).tolist():
        idxs = np.nonzero(lab == c)[0].tolist()
        rng.shuffle(idxs)
        n = len(idxs)
        if n <= 1:
            test_idx.extend(idxs)
            continue
        n_test = min(max(1, int(round(n * test_size))), n - 1)
        test_idx.extend(idxs[:n_test])
        train_idx.extend(idxs[n_test:])
    return train_idx, test_idx


def _cross_entropy(logits: torch.Tensor, y: torch.Tensor):
    """mean CE and softmax probabilities (log-sum-exp form)."""
    m = logits.max(dim=1, keepdim=True).values
    ex = torch.exp(logits - m)
    s = ex.sum(dim=1, keepdim=True)
    loss = (torch.log(s) + m - logits.gather(1, y.unsqueeze(1))).mean()
    return loss, ex / s


@torch.no_grad()
def train_linear_classifier(X_train, y_train, X_val, y_val, device, n_classes: int, max_steps: int = 1000,
                            lr: float = 1e-2, weight_decay: float = 0.0, patience: int = 20, min_delta: float = 0.0,
                            sync_every: int = 32, return_tensors: bool = False):
    """Full-batch linear probe: ``Linear(D, n_classes)`` + cross entropy, AdamW (betas 0.9/0.999, eps 1e-8), early
    stopping on the validation loss.  Inputs may be numpy arrays or tensors (tensors already on ``device`` are used
    in place).  Returns (best validation accuracy, y_true, y_pred) -- python lists like the reference, or device
    tensors with ``return_tensors=True``."""
    Xtr = torch.as_tensor(X_train, dtype=torch.float32).to(device)
    ytr = torch.as_tensor(y_train).to(device=device, dtype=torch.long)
    Xva = torch.as_tensor(X_val, dtype=torch.float32).to(device)
    yva = torch.as_tensor(y_val).to(device=device, dtype=torch.long)
    D = Xtr.shape[1]
    init = torch.nn.Linear(D, n_classes)               # the reference's initialisation (and its RNG consumption)
    W = init.weight.detach().to(device).clone()         # [C, D]
    b = init.bias.detach().to(device).clone()
    onehot = torch.zeros(Xtr.size(0), n_classes, device=device).scatter_(1, ytr.unsqueeze(1), 1.0)
    mW, vW, mb, vb = torch.zeros_like(W), torch.zeros_like(W), torch.zeros_like(b), torch.zeros_like(b)
    b1, b2, eps = 0.9, 0.999, 1e-8
    best_loss = torch.full((), float("inf"), device=device)
    best_W, best_b = W.clone(), b.clone()
    have_best = torch.zeros((), dtype=torch.bool, device=device)
    no_improve = torch.zeros((), dtype=torch.int32, device=device)
    inv_n = 1.0 / max(1, Xtr.size(0))
    for step in range(1, int(max_steps) + 1):
        active = no_improve < patience                  # device-side: once patience is exhausted nothing changes
        _, p = _cross_entropy(Xtr @ W.t() + b, ytr)
        g = (p - onehot) * inv_n                        # dL/dlogits of the mean cross entropy
        gW, gb = g.t() @ Xtr, g.sum(0)
        if weight_decay:
            W = torch.where(active, W * (1.0 - lr * weight_decay), W)
            b = torch.where(active, b * (1.0 - lr * weight_decay), b)
        mW = b1 * mW + (1 - b1) * gW
        vW = b2 * vW + (1 - b2) * gW * gW
        mb = b1 * mb + (1 - b1) * gb
        vb = b2 * vb + (1 - b2) * gb * gb
        c1, c2 = 1 - b1 ** step, 1 - b2 ** step
        W = torch.where(active, W - (lr / c1) * mW / (vW.sqrt() / (c2 ** 0.5) + eps), W)
        b = torch.where(active, b - (lr / c1) * mb / (vb.sqrt() / (c2 ** 0.5) + eps), b)
        v_loss, _ = _cross_entropy(Xva @ W.t() + b, yva)
        better = active & (v_loss < best_loss - min_delta)
        best_loss = torch.where(better, v_loss, best_loss)
        best_W = torch.where(better, W, best_W)
        best_b = torch.where(better, b, best_b)
        have_best = have_best | better
        no_improve = torch.where(better, torch.zeros_like(no_improve), no_improve + active.to(torch.int32))
        if step % sync_every == 0 and int(no_improve) >= patience:
            break
    W = torch.where(have_best, best_W, W)
    b = torch.where(have_best, best_b, b)
    preds = torch.argmax(Xva @ W.t() + b, dim=1)
    acc = (preds == yva).float().mean()
    if return_tensors:
        return acc, yva, preds
    return float(acc), yva.cpu().tolist(), preds.cpu().tolist()


def confusion_matrix(y_true: torch.Tensor, y_pred: torch.Tensor, n_classes: int) -> torch.Tensor:
    """[n_classes, n_classes] counts (rows = true, columns = predicted) with one bincount."""
    y_true, y_pred = torch.as_tensor(y_true).long(), torch.as_tensor(y_pred).long()
    ok = (y_true >= 0) & (y_true < n_classes) & (y_pred >= 0) & (y_pred < n_classes)
    return torch.bincount(y_true[ok] * n_classes + y_pred[ok], minlength=n_classes * n_classes).view(n_classes, n_classes)


def _prepare_targets(feats: dict, mkey: str, n_bins: int):
    y, edges = make_bin_labels(feats[mkey].to(torch.float32), n_bins=n_bins)
    return y.long(), n_bins, edges, _format_bin_names(edges, precision=4)


def _log(run, payload: dict):
    if run is not None and hasattr(run, "log"):
        run.log(payload)


def _run_probes(model, E: torch.Tensor, feats: dict, epoch: int, prefix: Optional[str], n_bins, test_size, steps, lr,
                rng_seed, patience, min_delta, save_csv) -> Dict[str, dict]:
    run = getattr(model, "wandb_run", None)
    targets = ["cum_area", "convex_hull", "labels"] + (["density"] if "density" in feats else [])
    results: Dict[str, dict] = {}
    for mkey in targets:
        if mkey not in feats:
            continue
        y, n_classes, edges, names = _prepare_targets(feats, mkey, n_bins)
        metric = f"{prefix}/{mkey}" if prefix else mkey
        tr, te = stratified_split(y, test_size=test_size, rng_seed=rng_seed)
        if not tr or not te:
            _log(run, {f"probe/{metric}/warn_empty_split/acc": 0.0, "epoch": epoch})
            continue
        tr_t = torch.as_tensor(tr, device=E.device)
        te_t = torch.as_tensor(te, device=E.device)
        acc, yt, yp = train_linear_classifier(E[tr_t], y[tr_t], E[te_t], y[te_t], device=E.device, n_classes=n_classes,
                                              max_steps=steps, lr=lr, weight_decay=0.0, patience=patience,
                                              min_delta=min_delta, return_tensors=True)
        cm = confusion_matrix(yt, yp, n_classes).cpu()
        results[metric] = {"acc": float(acc), "confusion": cm, "bin_names": names, "edges": edges.cpu()}
        _log(run, {f"probe/{metric}/acc": float(acc), "epoch": epoch})
        if save_csv:
            os.makedirs(model.arch_dir, exist_ok=True)
            path = os.path.join(model.arch_dir, f"probe_{metric.replace('/', '_')}_confusion_epoch{epoch}.csv")
            with open(path, "w") as f:
                f.write("True," + ",".join(names) + "\n")
                for name, row in zip(names, cm.tolist()):
                    f.write(name + "," + ",".join(str(int(v)) for v in row) + "\n")
            results[metric]["csv"] = path
    return results


def log_linear_probe(model, epoch: int, n_bins: int = 5, test_size: float = 0.2, steps: int = 1000, lr: float = 1e-2,
                     rng_seed: int = 42, patience: int = 20, min_delta: float = 0.0, save_csv: bool = True,
                     upto_layer: Optional[int] = None, layer_tag: Optional[str] = None) -> Dict[str, dict]:
    """Linear probes of the (image) stack's embeddings on the binned targets; returns {metric: {acc, confusion, ...}}
    (the reference only logs them)."""
    E, feats = compute_val_embeddings_and_features(model, upto_layer=upto_layer)
    return _run_probes(model, E, feats, epoch, layer_tag, n_bins, test_size, steps, lr, rng_seed, patience, min_delta, save_csv)


def log_joint_linear_probe(model, epoch: int, n_bins: int = 5, test_size: float = 0.2, steps: int = 1000,
                           lr: float = 1e-2, rng_seed: int = 42, patience: int = 20, min_delta: float = 0.0,
                           save_csv: bool = False, metric_prefix: str = "joint") -> Dict[str, dict]:
    """Linear probes of the joint-layer embeddings."""
    E, feats = compute_joint_embeddings_and_features(model)
    if E.numel() == 0:
        return {}
    return _run_probes(model, E, feats, epoch, metric_prefix, n_bins, test_size, steps, lr, rng_seed, patience, min_delta, save_csv)
