"""iDBN: stack of engine-backed RBMs with the reference's class surface.

Mirror of the reference ``imdbn/models/idbn.py`` for the parts on the hot path (SURVEY.md 8a:
a12-a14): constructor, the interleaved greedy ``train`` loop (:195-204), ``represent`` (:307-323),
``reconstruct`` (:325-344), ``decode`` (:346-359), ``save_model`` (:361-373).  The wandb / PCA /
linear-probe tail of ``train`` (:206-305) is the reference's observability side-car and is out of
scope; ``wandb_run`` is accepted and only receives the epoch loss.
"""
from __future__ import annotations

import os
import pickle
from typing import List, Optional

import torch

from imdbn.models.rbm import RBM
from imdbn.utils import batches, rows_on_device


class iDBN:
    def __init__(
        self,
        layer_sizes: List[int],
        params: dict,
        dataloader,
        val_loader,
        device,
        wandb_run=None,
        logging_config_path: Optional[str] = None,
    ):
        self.layers: List[RBM] = []
        self.params = params
        self.dataloader = dataloader
        self.val_loader = val_loader
        self.device = device
        self.wandb_run = wandb_run
        self.logging_cfg = {}

        # fields the reference's utilities expect (idbn.py:113-116)
        self.text_flag = False
        self.arch_str = "-".join(map(str, layer_sizes))
        self.arch_dir = os.path.join("logs-idbn", f"architecture_{self.arch_str}")
        os.makedirs(self.arch_dir, exist_ok=True)

        self.cd_k = int(self.params.get("CD", 1))                                   # idbn.py:118
        self.sparsity_last = bool(self.params.get("SPARSITY", False))
        self.sparsity_factor = float(self.params.get("SPARSITY_FACTOR", 0.1))

        try:                                                                        # idbn.py:123-126
            self.val_batch, self.val_labels = next(iter(val_loader))
        except Exception:
            self.val_batch, self.val_labels = None, None
        # validation features for the evaluation side-car (idbn.py:129-144): a Subset over a base dataset that carries
        # per-sample lists; any other loader leaves `features` at None, as in the reference
        self.features = None
        try:
            indices = val_loader.dataset.indices
            base = val_loader.dataset.dataset
            feats = {
                "Cumulative Area": torch.tensor([base.cumArea_list[i] for i in indices], dtype=torch.float32),
                "Convex Hull": torch.tensor([base.CH_list[i] for i in indices], dtype=torch.float32),
                "Labels": torch.tensor([base.labels[i] for i in indices], dtype=torch.float32),
            }
            density = getattr(base, "density_list", None)
            if density is not None:
                feats["Density"] = torch.tensor([density[i] for i in indices], dtype=torch.float32)
            self.features = feats
        except Exception:
            pass

        for i in range(len(layer_sizes) - 1):                                       # idbn.py:149-161
            rbm = RBM(
                num_visible=layer_sizes[i],
                num_hidden=layer_sizes[i + 1],
                learning_rate=self.params["LEARNING_RATE"],
                weight_decay=self.params["WEIGHT_PENALTY"],
                momentum=self.params["INIT_MOMENTUM"],
                dynamic_lr=self.params["LEARNING_RATE_DYNAMIC"],
                final_momentum=self.params["FINAL_MOMENTUM"],
                sparsity=(self.sparsity_last and i == len(layer_sizes) - 2),
                sparsity_factor=self.sparsity_factor,
            ).to(self.device)
            self.layers.append(rbm)

    def _layers_to_monitor(self) -> List[int]:
        layers = {len(self.layers)}
        if len(self.layers) > 1:
            layers.add(1)
        return sorted(layers)

    def _layer_tag(self, idx: int) -> str:
        return f"layer{idx}"

    def train(self, epochs: int, log_every_pca: int = 25, log_every_probe: int = 10):
        """Interleaved greedy layer-wise training (idbn.py:195-204): on every batch each layer
        is updated by CD and then feeds the next layer with p(h|v) from its UPDATED weights.

        The reference does ``float(loss)`` after every update (a device->host sync per RBM
        update, idbn.py:204); here losses stay on the device and are fetched once per epoch
        (values identical).  ``self.loss_history`` keeps them.
        """
        self.loss_history = []
        for epoch in range(int(epochs)):
            losses = []
            # one batch of lookahead: the first layer prepares the operand forms of the following batch during its
            # weight update (RBM.train_epoch next_data=); the batches and their order are unchanged
            dev_batch = lambda item: None if item is None else rows_on_device(item[0], self.device)
            it = iter(batches(self.dataloader))
            cur = dev_batch(next(it, None))
            while cur is not None:
                nxt = dev_batch(next(it, None))
                v = cur
                last = len(self.layers) - 1
                for li, rbm in enumerate(self.layers):
                    # update + forward of the same batch as one engine call; the top layer's forward (computed and
                    # dropped by the reference, idbn.py:203) has no side effect and is not run
                    if li < last:
                        loss, v = rbm.train_epoch(v, epoch, epochs, CD=self.cd_k, next_data=nxt if li == 0 else None, return_forward=True)
                    else:
                        loss = rbm.train_epoch(v, epoch, epochs, CD=self.cd_k, next_data=nxt if li == 0 else None)
                    losses.append(loss)
                cur = nxt
            if losses:
                ep = torch.stack([l.reshape(()) for l in losses]).float().cpu()
                self.loss_history.append(ep)
                if self.wandb_run:
                    self.wandb_run.log({"idbn/loss": float(ep.mean()), "epoch": epoch})

    @torch.no_grad()
    def represent(self, x: torch.Tensor, upto_layer: Optional[int] = None) -> torch.Tensor:
        """idbn.py:319-323."""
        v = rows_on_device(x, self.device)
        L = len(self.layers) if (upto_layer is None) else max(0, min(len(self.layers), int(upto_layer)))
        for i in range(L):
            v = self.layers[i].forward(v)
        return v

    @torch.no_grad()
    def reconstruct(self, x: torch.Tensor) -> torch.Tensor:
        """idbn.py:336-344."""
        cur = rows_on_device(x, self.device)
        for rbm in self.layers:
            cur = rbm.forward(cur)
        for rbm in reversed(self.layers):
            cur = rbm.backward(cur)
        return cur

    @torch.no_grad()
    def decode(self, top: torch.Tensor) -> torch.Tensor:
        """idbn.py:356-359."""
        cur = top.to(self.device)
        for rbm in reversed(self.layers):
            cur = rbm.backward(cur)
        return cur

    def save_model(self, path: str):
        """idbn.py:370-372: pickle of {"layers", "params"} (live RBM modules)."""
        model_copy = {"layers": self.layers, "params": self.params}
        with open(path, "wb") as f:
            pickle.dump(model_copy, f)
        print(f"[iDBN] Model saved to {path}")
