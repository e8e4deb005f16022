"""iMDBN_BiModal: two modality iDBNs joined by a (multi-layer) joint DBN over [z_mod1 | z_mod2], on the MI355X engine.

Mirror of the reference ``imdbn/models/imdbn_bimodal.py`` for the training / inference path:
constructor and ``_build_joint`` (:437-575), ``load_pretrained_mod{1,2}_dbn`` (:577-615),
``init_joint_bias_from_data`` (:617-645), ``_cross_reconstruct`` (:648-693), ``represent`` (:696-709), the
``train_joint`` batch loop with its online cross-modal MSE (:711-826), ``save_model`` / ``load_model`` (:1017-1076).
The wandb / PCA / probe / trajectory / snapshot tail (:828-1015) is the observability side-car and is out of scope.

Data parallelism (``imdbn.engine.dp``): the RBM updates shard by rows themselves; the two host-side accumulators here (the
bias-initialisation counters and the per-epoch cross-modal MSE sums) are all-reduced once each, as in ``iMDBN``.

Only orchestration lives here: every RBM operation (CD-k, clamped CD-3 with sampled hidden units, the noisy
mean-field and Gibbs chains, propagations) is one engine call.
"""
from __future__ import annotations

import datetime
import pickle
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from imdbn import engine as _E
from imdbn.models.idbn import iDBN
from imdbn.models.rbm import RBM
from imdbn.utils import batches, rows_on_device

WARMUP_EPOCHS = 8            # imdbn_bimodal.py:736
AUX_CD = 3                   # :762,:775,:800,:814  (clamped updates run CD-3 with sampled hidden units)


class iMDBN_BiModal(nn.Module):
    def __init__(self, layer_sizes_mod1: list, layer_sizes_mod2: list, joint_layer_sizes, params: Optional[dict] = None,
                 dataloader=None, val_loader=None, device=None, wandb_run=None, logging_cfg: Optional[dict] = None):
        super().__init__()
        self.params = params or {}
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.dataloader = dataloader
        self.val_loader = val_loader
        self.wandb_run = wandb_run
        self.logging_cfg = logging_cfg or {}

        self.mod1_dbn = iDBN(layer_sizes=layer_sizes_mod1, params=self.params, dataloader=None, val_loader=None,
                             device=self.device, wandb_run=self.wandb_run)                      # :469-476
        self.mod2_dbn = iDBN(layer_sizes=layer_sizes_mod2, params=self.params, dataloader=None, val_loader=None,
                             device=self.device, wandb_run=self.wandb_run)                      # :479-486
        self.Dz_mod1 = int(self.mod1_dbn.layers[-1].num_hidden)
        self.Dz_mod2 = int(self.mod2_dbn.layers[-1].num_hidden)
        self._build_joint(joint_layer_sizes)

        self.joint_cd = int(self.params.get("JOINT_CD", self.params.get("CD", 1)))              # :496-497
        self.cross_steps = int(self.params.get("CROSS_GIBBS_STEPS", 50))

        try:                                                                                    # :500-506
            vb_mod1, vb_mod2 = next(iter(val_loader))
            self.validation_mod1 = vb_mod1[:8].to(self.device)
            self.validation_mod2 = vb_mod2[:8].to(self.device)
        except Exception:
            self.validation_mod1 = None
            self.validation_mod2 = None

        self.features = None                                                                    # :509-535
        try:
            if hasattr(val_loader.dataset, "indices"):
                indices, base = val_loader.dataset.indices, val_loader.dataset.dataset
            else:
                base = val_loader.dataset
                indices = range(len(base))
            self.features = {
                "Cumulative Area": torch.tensor([base.cumArea_list[i] for i in indices], dtype=torch.float32),
                "Convex Hull": torch.tensor([base.CH_list[i] for i in indices], dtype=torch.float32),
                "Labels": torch.tensor([base.labels[i] for i in indices], dtype=torch.float32),
            }
            density = getattr(base, "density_list", None)
            if density is not None:
                self.features["Density"] = torch.tensor([density[i] for i in indices], dtype=torch.float32)
        except Exception:
            self.features = None

        joint_for_str = joint_layer_sizes if isinstance(joint_layer_sizes, list) else [joint_layer_sizes]
        self.arch_str = (f"MOD1{'-'.join(map(str, layer_sizes_mod1))}_MOD2{'-'.join(map(str, layer_sizes_mod2))}"
                         f"_JOINT{'-'.join(map(str, joint_for_str))}")
        self.joint_history = []

    def _build_joint(self, joint_layer_sizes):
        """:543-575 -- a stack of plain (no softmax group) RBMs over [z_mod1 | z_mod2]."""
        if isinstance(joint_layer_sizes, int):
            joint_layer_sizes = [joint_layer_sizes]
        p = self.params
        self.joint_layers = []
        visible = self.Dz_mod1 + self.Dz_mod2
        for hidden in joint_layer_sizes:
            self.joint_layers.append(RBM(
                num_visible=visible, num_hidden=int(hidden),
                learning_rate=p.get("JOINT_LEARNING_RATE", p.get("LEARNING_RATE", 0.1)),
                weight_decay=p.get("WEIGHT_PENALTY", 0.0001), momentum=p.get("INIT_MOMENTUM", 0.5),
                dynamic_lr=p.get("LEARNING_RATE_DYNAMIC", True), final_momentum=p.get("FINAL_MOMENTUM", 0.95),
                softmax_groups=[],
            ).to(self.device))
            visible = int(hidden)
        self.joint_rbm = self.joint_layers[0]
        self.num_joint_layers = len(self.joint_layers)

    # ---- pretrained modality stacks (:577-615) -------------------------------------------------
    def load_pretrained_mod1_dbn(self, path: str) -> bool:
        return self._load_pretrained_dbn(self.mod1_dbn, path, "mod1")

    def load_pretrained_mod2_dbn(self, path: str) -> bool:
        return self._load_pretrained_dbn(self.mod2_dbn, path, "mod2")

    def _load_pretrained_dbn(self, dbn: iDBN, path: str, name: str) -> bool:
        try:
            with open(path, "rb") as f:
                obj = pickle.load(f)
        except Exception as e:
            print(f"[load_pretrained_{name}_dbn] error: {e}")
            return False
        if isinstance(obj, dict) and "layers" in obj:
            dbn.layers = obj["layers"]
        elif hasattr(obj, "layers"):
            dbn.layers = obj.layers
        else:
            print(f"[load_pretrained_{name}_dbn] unrecognized format")
            return False
        for rbm in dbn.layers:                      # re-home on the device, momentum re-zeroed (:603-612)
            rbm.to(self.device)
            rbm.W_m = torch.zeros_like(rbm.W)
            rbm.hb_m = torch.zeros_like(rbm.hid_bias)
            rbm.vb_m = torch.zeros_like(rbm.vis_bias)
            if not hasattr(rbm, "softmax_groups"):
                rbm.softmax_groups = []
        print(f"[load_pretrained_{name}_dbn] loaded from {path}")
        return True

    # ---- bias initialisation (:617-645) --------------------------------------------------------
    @torch.no_grad()
    def init_joint_bias_from_data(self, n_batches: int = 10):
        sum_z1 = sum_z2 = None
        n = 0
        for b, (mod1, mod2) in enumerate(batches(self.dataloader)):
            if b >= n_batches:
                break
            z1 = self.mod1_dbn.represent(rows_on_device(mod1, self.device))
            z2 = self.mod2_dbn.represent(rows_on_device(mod2, self.device))
            sum_z1 = z1.sum(0) if sum_z1 is None else (sum_z1 + z1.sum(0))
            sum_z2 = z2.sum(0) if sum_z2 is None else (sum_z2 + z2.sum(0))
            n += z1.size(0)
        if _E.dp.active():
            # every rank saw its own shard of the first batches: the counters are sums over rows
            D1, D2 = self.Dz_mod1, self.Dz_mod2
            pack = torch.cat([(sum_z1 if sum_z1 is not None else torch.zeros(D1, device=self.device)).double(),
                              (sum_z2 if sum_z2 is not None else torch.zeros(D2, device=self.device)).double(),
                              torch.tensor([float(n)], device=self.device, dtype=torch.float64)])
            _E.dp.all_reduce_sum(pack)
            sum_z1, sum_z2, n = pack[:D1].float(), pack[D1:D1 + D2].float(), int(round(float(pack[D1 + D2])))
        if n == 0:
            return
        mean_z1 = (sum_z1 / n).clamp(1e-4, 1 - 1e-4)
        mean_z2 = (sum_z2 / n).clamp(1e-4, 1 - 1e-4)
        vb = self.joint_layers[0].vis_bias
        vb.data[: self.Dz_mod1] = torch.log(mean_z1) - torch.log1p(-mean_z1)
        vb.data[self.Dz_mod1:] = torch.log(mean_z2) - torch.log1p(-mean_z2)

    # ---- inference (:648-709) --------------------------------------------------------------------
    def _clamp(self, z, lo: int, B: int):
        V = self.Dz_mod1 + self.Dz_mod2
        vk = torch.zeros(B, V, device=self.device)
        km = torch.zeros(B, V, device=self.device)
        vk[:, lo:lo + z.size(1)] = z
        km[:, lo:lo + z.size(1)] = 1.0
        return vk, km

    @torch.no_grad()
    def _cross_reconstruct(self, z_mod1: torch.Tensor, z_mod2: torch.Tensor,
                           steps: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Returns (mod1_from_mod2, mod2_from_mod1): Gibbs completion with sampled hidden units in both
        directions, then decoding through the modality stacks."""
        if steps is None:
            steps = self.cross_steps
        B, Dz1 = z_mod1.size(0), self.Dz_mod1
        vk, km = self._clamp(z_mod1, 0, B)
        v12 = self.joint_rbm.conditional_gibbs(vk, km, n_steps=steps, sample_h=True, sample_v=False)
        vk, km = self._clamp(z_mod2, Dz1, B)
        v21 = self.joint_rbm.conditional_gibbs(vk, km, n_steps=steps, sample_h=True, sample_v=False)
        return self.mod1_dbn.decode(v21[:, :Dz1]), self.mod2_dbn.decode(v12[:, Dz1:])

    @torch.no_grad()
    def represent(self, batch: Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
        mod1, mod2 = batch
        z1 = self.mod1_dbn.represent(rows_on_device(mod1, self.device))
        z2 = self.mod2_dbn.represent(rows_on_device(mod2, self.device))
        h = torch.cat([z1, z2], dim=1)
        for rbm in self.joint_layers:
            h = rbm.forward(h)
        return h

    # ---- joint training (:711-826) ---------------------------------------------------------------
    def train_joint(self, epochs: int, log_every: int = 5, log_every_pca: int = 25, log_every_probe: int = 10,
                    log_every_trajectory: int = 50):
        """Warm-up (epochs < 8): per batch 2x (mod1-clamped, mod2-clamped) CD-3 updates of the first joint layer;
        then: free CD through all joint layers (each trained on the previous one's activations) + one mod1- and
        one mod2-clamped CD-3 update of the first layer.  Cross-modal MSE on every batch; the accumulators stay
        on the device and are fetched once per epoch (``self.joint_history``)."""
        print(f"[iMDBN_BiModal] joint training: {self.num_joint_layers} layers, {epochs} epochs total")
        self.init_joint_bias_from_data(n_batches=10)
        Dz1 = self.Dz_mod1
        aux_steps = int(self.params.get("JOINT_AUX_COND_STEPS", 30))
        first = self.joint_layers[0]
        self.joint_history = []
        for epoch in range(int(epochs)):
            cd_losses = []
            acc = torch.zeros(3, device=self.device, dtype=torch.float64)      # n, mse_mod1_sum, mse_mod2_sum
            for mod1, mod2 in batches(self.dataloader):
                v1 = rows_on_device(mod1, self.device)
                v2 = rows_on_device(mod2, self.device)
                B = v1.size(0)
                with torch.no_grad():
                    z1 = self.mod1_dbn.represent(v1)
                    z2 = self.mod2_dbn.represent(v2)
                if epoch < WARMUP_EPOCHS:                                                       # :750-781
                    for _ in range(2):
                        for z, lo in ((z1, 0), (z2, Dz1)):
                            vk, km = self._clamp(z, lo, B)
                            first.train_epoch_clamped(vk, km, epoch, epochs, CD=AUX_CD, cond_init_steps=aux_steps,
                                                      sample_h=True, sample_v=False, aux_lr_mult=0.3, use_noisy_init=True)
                else:                                                                           # :783-820
                    cur = torch.cat([z1, z2], dim=1)
                    for li, rbm in enumerate(self.joint_layers):
                        loss = rbm.train_epoch(cur, epoch, epochs, CD=self.joint_cd)
                        if li == 0:
                            cd_losses.append(loss)
                        cur = rbm.forward(cur)
                    for z, lo in ((z1, 0), (z2, Dz1)):
                        vk, km = self._clamp(z, lo, B)
                        first.train_epoch_clamped(vk, km, epoch, epochs, CD=AUX_CD, cond_init_steps=aux_steps,
                                                  sample_h=True, sample_v=False, reclamp_negative=False,
                                                  aux_lr_mult=0.3, use_noisy_init=True)
                with torch.no_grad():                                                           # :823-829
                    r1, r2 = self._cross_reconstruct(z1, z2, steps=self.cross_steps)
                    acc[0] += B
                    acc[1] += F.mse_loss(r1.view_as(v1), v1, reduction="sum").double()
                    acc[2] += F.mse_loss(r2.view_as(v2), v2, reduction="sum").double()
            if _E.dp.active():                      # each rank accumulated its shard: sums over rows
                _E.dp.all_reduce_sum(acc)
            n, s1, s2 = (float(x) for x in acc.tolist())                            # one host sync per epoch
            npix1, npix2 = self.mod1_dbn.layers[0].num_visible, self.mod2_dbn.layers[0].num_visible
            self.joint_history.append({
                "epoch": epoch,
                "cd_losses": torch.stack([l.detach().reshape(()) for l in cd_losses]).cpu() if cd_losses else None,
                "mod1_mse": s1 / (n * npix1) if n else None,
                "mod2_mse": s2 / (n * npix2) if n else None,
            })
        print("[iMDBN_BiModal] joint training finished.")

    # ---- persistence (:1017-1076) ------------------------------------------------------------------
    @torch.no_grad()
    def save_model(self, path: str):
        payload = {
            "mod1_dbn": self.mod1_dbn, "mod2_dbn": self.mod2_dbn, "joint_layers": self.joint_layers,
            "num_joint_layers": self.num_joint_layers, "Dz_mod1": self.Dz_mod1, "Dz_mod2": self.Dz_mod2,
            "params": self.params, "arch_str": self.arch_str, "features": self.features,
            "metadata": {"saved_at": datetime.datetime.now().isoformat(), "model_type": "iMDBN_BiModal",
                         "architecture": self.arch_str},
        }
        with open(path, "wb") as f:
            pickle.dump(payload, f)
        print(f"[iMDBN_BiModal] Model saved to {path}")

    @staticmethod
    def load_model(path: str, device=None) -> Dict[str, Any]:
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        with open(path, "rb") as f:
            payload = pickle.load(f)
        for key in ("mod1_dbn", "mod2_dbn"):
            if key in payload:
                for rbm in payload[key].layers:
                    rbm.to(device)
        if "joint_layers" in payload:
            for rbm in payload["joint_layers"]:
                rbm.to(device)
        elif "joint_rbm" in payload:                # older single-RBM files (:1062-1066)
            payload["joint_rbm"].to(device)
            payload["joint_layers"] = [payload["joint_rbm"]]
            payload["num_joint_layers"] = 1
        print(f"[iMDBN_BiModal] Model loaded from {path}")
        return payload
