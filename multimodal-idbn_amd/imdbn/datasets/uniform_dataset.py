"""Stimulus dataset + device-resident loaders.

The reference imports ``imdbn.datasets.uniform_dataset.create_dataloaders_uniform`` (``scripts/train_multimodal.py:11,
96-102``, ``README.md:44-52``) but does not ship the module; what it fixes is the *attribute contract* the models read
(``idbn.py:131-137``): ``loader.dataset`` is a ``Subset`` -- ``.indices`` into ``.dataset`` -- over a base dataset
with per-sample lists ``labels``, ``cumArea_list``, ``CH_list`` and optionally ``density_list``; a batch is
``(images [B, 10000] or [B, 100, 100], labels)`` with one-hot labels in the multimodal case (``imdbn.py:560-562``).

MI355X-first layout: with the engine a batch-64 update of the 10000 <-> 1500 layer takes 0.13 ms, so a per-batch
host collate + H2D copy of 64 x 10000 fp32 (2.6 MB) would dominate.  The whole split therefore lives in HBM once
(uint8 pixels: 10 KB per stimulus, 100 k stimuli = 1 GB of the 288 GB) and ``DeviceLoader`` serves batches as device
tensors: one ``randperm`` per epoch, one gather + uint8 -> fp32 expansion per batch, no host work in the loop.

``.npz`` keys (aliases accepted): images ``D`` | ``images`` | ``X`` ([N, 10000] or [N, 100, 100], any numeric dtype,
values 0/1 or 0..255), numerosity ``N_list`` | ``labels`` | ``y``, cumulative area ``cumArea_list`` | ``cum_area``,
convex hull ``CH_list`` | ``convex_hull``, density ``density_list`` | ``density`` (optional).
"""
from __future__ import annotations

import os
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset, Subset

_ALIASES = {
    "images": ("D", "images", "X", "data"),
    "labels": ("N_list", "labels", "y", "numerosity"),
    "cumArea_list": ("cumArea_list", "cum_area", "cumulative_area", "CA_list"),
    "CH_list": ("CH_list", "convex_hull", "ch"),
    "density_list": ("density_list", "density", "Density"),
}


def _pick(z, field: str, required: bool):
    for k in _ALIASES[field]:
        if k in z:
            return np.asarray(z[k])
    if required:
        raise KeyError(f"{field}: none of {_ALIASES[field]} in the archive (has {sorted(z.keys())})")
    return None


class UniformDataset(Dataset):
    """In-memory stimulus set with the attribute contract of idbn.py:131-137.

    ``labels`` are the raw numerosities; ``classes`` the sorted distinct values; ``label_index[i]`` the class index of
    sample i.  ``__getitem__`` returns ``(image fp32 [P], one-hot [K])`` when ``multimodal_flag`` else
    ``(image, class index)``.
    """

    def __init__(self, path2data: Optional[str] = None, data_name: Optional[str] = None, multimodal_flag: bool = True,
                 arrays: Optional[Dict[str, np.ndarray]] = None):
        if arrays is None:
            path = os.path.join(path2data or "", data_name or "")
            if not os.path.isfile(path):
                raise FileNotFoundError(f"dataset archive not found: {path}")
            with np.load(path, allow_pickle=False) as z:
                arrays = {k: z[k] for k in z.files}
        img = _pick(arrays, "images", True)
        img = img.reshape(img.shape[0], -1)
        if img.dtype != np.uint8:
            mx = float(img.max()) if img.size else 0.0
            img = (img > 0.5 * mx).astype(np.uint8) * np.uint8(255) if mx > 1.0 else np.rint(np.clip(img, 0, 1) * 255).astype(np.uint8)
        elif img.size and img.max() <= 1:
            img = img * np.uint8(255)
        self.images_u8 = torch.from_numpy(np.ascontiguousarray(img))            # [N, P], 0..255 (255 = 1.0)
        lab = _pick(arrays, "labels", True).reshape(-1)
        n = self.images_u8.size(0)
        if lab.shape[0] != n:
            raise ValueError(f"{lab.shape[0]} labels for {n} images")
        self.labels = lab.tolist()
        self.classes = sorted(set(self.labels))
        lut = {c: i for i, c in enumerate(self.classes)}
        self.label_index = torch.tensor([lut[c] for c in self.labels], dtype=torch.long)
        self.num_classes = len(self.classes)
        for field in ("cumArea_list", "CH_list", "density_list"):
            v = _pick(arrays, field, False)
            if v is not None and v.reshape(-1).shape[0] != n:
                raise ValueError(f"{field}: {v.reshape(-1).shape[0]} values for {n} images")
            if v is None and field != "density_list":
                v = np.zeros(n, np.float32)
            setattr(self, field, None if v is None else v.reshape(-1).astype(np.float32).tolist())
        self.multimodal_flag = bool(multimodal_flag)

    def __len__(self) -> int:
        return self.images_u8.size(0)

    def __getitem__(self, i: int):
        x = self.images_u8[i].float() / 255.0
        k = self.label_index[i]
        if self.multimodal_flag:
            return x, torch.nn.functional.one_hot(k, self.num_classes).float()
        return x, k


class DeviceLoader:
    """Batches of a ``Subset`` of a ``UniformDataset`` served from device memory.

    Iterating yields ``(images fp32 [b, P], labels)`` tensors on ``device`` -- ``labels`` one-hot fp32 [b, K] for a
    multimodal base, class indices otherwise.  ``shuffle=True`` draws one device ``randperm`` per epoch from its own
    generator (seeded; the epoch order is reproducible and independent of the model's draws).  Exposes what the models
    and the side-car read from a DataLoader: ``dataset`` (the Subset: ``.indices``, ``.dataset``), ``batch_size``,
    ``__len__``; under data parallelism pass ``rank`` / ``world_size`` and every rank serves rows
    ``[rank*b, (rank+1)*b)`` of each global batch of ``world_size * batch_size`` rows.
    """

    def __init__(self, subset: Subset, batch_size: int, device, shuffle: bool = False, drop_last: bool = False,
                 seed: int = 0, rank: int = 0, world_size: int = 1):
        base = subset.dataset
        self.dataset = subset
        self.batch_size = int(batch_size)
        self.device = torch.device(device)
        self.shuffle, self.drop_last = bool(shuffle), bool(drop_last)
        self.rank, self.world_size = int(rank), int(world_size)
        idx = torch.as_tensor(list(subset.indices), dtype=torch.long)
        self._x = base.images_u8[idx].to(self.device)                              # [n, P] uint8, resident
        self._k = base.label_index[idx].to(self.device)
        self._K = base.num_classes
        self._onehot = base.multimodal_flag
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(int(seed))
        self.epoch = 0
        # binary stimuli (every pixel 0 or 255): the batches carry the tag the engine reads instead of checking each one
        # (HipEngine.binary_hint): their CD updates read the images as bit planes
        self._binary = bool(((self._x == 0) | (self._x == 255)).all().item()) if self._x.numel() else False

    def _global(self) -> int:
        return self.batch_size * self.world_size

    def __len__(self) -> int:
        n, g = self._x.size(0), self._global()
        return n // g if self.drop_last else (n + g - 1) // g

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        n, g, b = self._x.size(0), self._global(), self.batch_size
        order = torch.randperm(n, device=self.device, generator=self._gen) if self.shuffle else None
        self.epoch += 1
        for i in range(len(self)):
            lo = i * g + self.rank * b
            hi = min(lo + b, min((i + 1) * g, n))
            if hi <= lo:
                continue
            if order is None:
                x, k = self._x[lo:hi], self._k[lo:hi]
            else:
                sel = order[lo:hi]
                x, k = self._x.index_select(0, sel), self._k.index_select(0, sel)
            xf = x.to(torch.float32).mul_(1.0 / 255.0)
            xf._imdbn_binary = self._binary
            yield xf, (torch.nn.functional.one_hot(k, self._K).to(torch.float32) if self._onehot else k)


def _stratified_indices(label_index: torch.Tensor, fractions: Sequence[float], seed: int) -> List[List[int]]:
    """Per-class shuffled split: every class is represented in every part in the same proportions ("uniform")."""
    g = np.random.Generator(np.random.PCG64(seed))
    lab = label_index.numpy()
    parts: List[List[int]] = [[] for _ in fractions]
    for c in np.unique(lab):
        idx = np.nonzero(lab == c)[0]
        g.shuffle(idx)
        n = len(idx)
        cuts, acc = [], 0
        for f in fractions[1:]:
            m = int(round(n * f)) if n > 1 else 0
            cuts.append(m)
            acc += m
        acc = min(acc, max(0, n - 1))                    # leave at least one sample of the class in train
        pos = n
        for j in range(len(fractions) - 1, 0, -1):
            m = min(cuts[j - 1], max(0, pos - 1))
            parts[j].extend(idx[pos - m:pos].tolist())
            pos -= m
        parts[0].extend(idx[:pos].tolist())
    return [sorted(p) for p in parts]


def create_dataloaders_uniform(path2data: Optional[str] = None, data_name: Optional[str] = None, batch_size: int = 64,
                               num_workers: int = 1, multimodal_flag: bool = True, val_size: float = 0.1,
                               test_size: float = 0.1, data_path: Optional[str] = None, device=None, seed: int = 0,
                               shuffle_train: bool = True, rank: int = 0, world_size: int = 1,
                               dataset: Optional[UniformDataset] = None):
    """``(train_loader, val_loader, test_loader)`` over a class-stratified split of the stimulus archive.

    Keyword names of both call styles in the reference are accepted (``path2data=`` README.md:48-52, ``data_path=``
    scripts/train_multimodal.py:96-102); ``num_workers`` is accepted and unused (there is no host work per batch).
    ``device`` defaults to the current accelerator.  Validation / test loaders are sequential, so
    ``idbn.py:131-137``-style feature extraction lines up with the embeddings of ``probe_utils``.
    """
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    base = dataset if dataset is not None else UniformDataset(path2data if path2data is not None else data_path,
                                                              data_name, multimodal_flag)
    tr, va, te = _stratified_indices(base.label_index, (1.0 - val_size - test_size, val_size, test_size), seed)
    mk = lambda idx, sh, r, w: DeviceLoader(Subset(base, idx), batch_size, device, shuffle=sh, seed=seed + 1, rank=r, world_size=w)
    return mk(tr, shuffle_train, rank, world_size), mk(va, False, 0, 1), mk(te, False, 0, 1)
