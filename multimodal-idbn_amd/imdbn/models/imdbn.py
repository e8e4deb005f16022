"""iMDBN: image iDBN + joint RBM over [z_img, y_onehot], backed by the MI355X CD engine.

Mirror of the reference ``imdbn/models/imdbn.py`` for the hot path (a15-a21):
constructor / ``_build_joint`` (:68-214), ``init_joint_bias_from_data`` (:216-292),
``load_pretrained_image_idbn`` (:294-342), ``finetune_image_last_layer`` (:344-384),
``_cross_reconstruct`` (:386-488), ``represent`` (:490-506), the ``train_joint`` batch loop with its
online metrics (:553-639), ``save_model`` / ``load_model`` (:815-934).  The wandb / PCA / probe /
snapshot tail (:641-813) is the observability side-car and is out of scope.

All RBM arithmetic goes through the engine; the few torch ops left here (concatenation, class-mean
bookkeeping, argmax/top-k of the online metrics) are host-logic plumbing on small tensors.
"""
from __future__ import annotations

import datetime
import os
import pickle
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from imdbn import engine as _E
from imdbn.models.idbn import iDBN
from imdbn.models.rbm import RBM
from imdbn.utils import batches, rows_on_device

WARMUP_Y_EPOCHS = 8          # imdbn.py:540
Z_CLAMP_EVERY = 50           # imdbn.py:600
KBUF = 5                     # imdbn.py:452


class _MetricsOverlap:
    """train_joint's metrics pass on a second stream (see iMDBN.train_joint).  Two snapshots of the joint RBM alternate, so the main
    stream never waits for the metrics of the batch before; it waits (normally: not at all) for those of two batches before."""

    def __init__(self, model):
        self.m = model
        jr = model.joint_rbm
        dev = jr.W.device
        self.main = torch.cuda.current_stream(dev)
        self.side = torch.cuda.Stream(dev)
        self.snaps = []
        with torch.random.fork_rng(devices=[dev]):             # (the constructor draws initial weights: leave torch's generators as they were)
            for _ in range(2):
                r = RBM(jr.num_visible, jr.num_hidden, jr.lr, jr.weight_decay, jr.momentum, dynamic_lr=jr.dynamic_lr,
                        final_momentum=jr.final_momentum, softmax_groups=list(jr.softmax_groups)).to(dev)
                self.snaps.append(r)
        self.done = [None, None]
        self.n = 0

    def submit(self, acc, z_img, y, img):
        k = self.n & 1
        self.n += 1
        jr, snap = self.m.joint_rbm, self.snaps[k]
        if self.done[k] is not None:
            self.main.wait_event(self.done[k])                 # the metrics that read this snapshot two batches ago are through
        snap.W.data.copy_(jr.W.data); snap.hid_bias.data.copy_(jr.hid_bias.data); snap.vis_bias.data.copy_(jr.vis_bias.data)
        ready = torch.cuda.Event()
        ready.record(self.main)
        for t in (acc, z_img, y, img):
            t.record_stream(self.side)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            self.m._batch_metrics(acc, snap, z_img, y, img)
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.done[k] = ev

    def join(self):
        self.main.wait_stream(self.side)


class iMDBN(nn.Module):
    def __init__(
        self,
        layer_sizes_img: list,
        layer_sizes_txt_or_joint=None,
        joint_layer_size: Optional[int] = None,
        params: Optional[dict] = None,
        dataloader=None,
        val_loader=None,
        device=None,
        text_posenc_dim: int = 0,
        num_labels: int = 32,
        embedding_dim: int = 64,
        wandb_run=None,
        logging_config_path: Optional[str] = None,
        logging_cfg: Optional[dict] = None,
    ):
        super().__init__()
        # two accepted signatures (imdbn.py:105-112): (img_layers, txt_layers, joint) or (img_layers, joint)
        if isinstance(layer_sizes_txt_or_joint, (list, tuple)):
            if joint_layer_size is None:
                raise ValueError("joint_layer_size required with legacy constructor signature")
        elif joint_layer_size is None:
            joint_layer_size = int(layer_sizes_txt_or_joint)

        self.params = params or {}
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.dataloader = dataloader
        self.val_loader = val_loader
        self.wandb_run = wandb_run
        self.logging_cfg = logging_cfg if logging_cfg is not None else {}
        self.num_labels = int(num_labels)

        try:                                                                        # imdbn.py:137-145
            vb_imgs, vb_lbls = next(iter(val_loader))
            self.validation_images = vb_imgs[:8].to(self.device)
            self.validation_labels = vb_lbls[:8].to(self.device)
            self.val_batch = (vb_imgs, vb_lbls)
        except Exception:
            self.validation_images = None
            self.validation_labels = None
            self.val_batch = None

        self.image_idbn = iDBN(
            layer_sizes=layer_sizes_img, params=self.params, dataloader=self.dataloader,
            val_loader=self.val_loader, device=self.device, wandb_run=self.wandb_run,
            logging_config_path=logging_config_path,
        )
        self.Dz_img = int(self.image_idbn.layers[-1].num_hidden)
        self._build_joint(Dz_img=self.Dz_img, joint_hidden=joint_layer_size)

        self.joint_cd = int(self.params.get("JOINT_CD", self.params.get("CD", 1)))  # imdbn.py:164-167
        self.cross_steps = int(self.params.get("CROSS_GIBBS_STEPS", 50))
        # Best-of-K refinement of the TXT->IMG chain (imdbn.py:451-474).  The reference's selection is inert (no
        # RBM.free_energy -> all energies 0 -> candidate 0); that stays the default.  Opt in with
        # CROSS_LIVE_BEST_OF_K=True (K from CROSS_BEST_OF_K, reference buffer size 5): the K-1 one-step refinement
        # passes are computed, every candidate gets its free energy and each row keeps its lowest-energy candidate.
        self.live_best_of_k = bool(self.params.get("CROSS_LIVE_BEST_OF_K", False))
        self.best_of_k = int(self.params.get("CROSS_BEST_OF_K", KBUF))
        self.aux_every_k = int(self.params.get("JOINT_AUX_EVERY_K", 0))
        self.aux_cond_steps = int(self.params.get("JOINT_AUX_COND_STEPS", 50))
        # imdbn.py:170-187: same extraction as the image stack did from the same val_loader
        self.features = getattr(self.image_idbn, "features", None)
        self.arch_str = f"IMG{'-'.join(map(str, layer_sizes_img))}_JOINT{joint_layer_size}"

    def _build_joint(self, Dz_img: int, joint_hidden: int):
        """imdbn.py:203-214: visible = [z_img (Dz) | y (K, one softmax group)]."""
        self.Dz_img = int(Dz_img)
        K = self.num_labels
        p = self.params
        self.joint_rbm = RBM(
            num_visible=self.Dz_img + K,
            num_hidden=int(joint_hidden),
            learning_rate=p.get("JOINT_LEARNING_RATE", p.get("LEARNING_RATE", 0.1)),
            weight_decay=p.get("WEIGHT_PENALTY", 0.0001),
            momentum=p.get("INIT_MOMENTUM", 0.5),
            dynamic_lr=p.get("LEARNING_RATE_DYNAMIC", True),
            final_momentum=p.get("FINAL_MOMENTUM", 0.95),
            softmax_groups=[(self.Dz_img, self.Dz_img + K)],
        ).to(self.device)

    # ---- bias initialisation (imdbn.py:216-292) ---------------------------------------------------
    @torch.no_grad()
    def init_joint_bias_from_data(self, n_batches: int = 10):
        if not hasattr(self, "Dz_img"):
            self.Dz_img = int(self.joint_rbm.num_visible) - self.num_labels
        Dz, K = self.Dz_img, self.num_labels
        sum_z, n = None, 0
        class_counts = torch.zeros(K, device=self.device)
        zs, ys = [], []
        for b, (imgs, lbls) in enumerate(batches(self.dataloader)):
            if b >= n_batches:
                break
            z = self.image_idbn.represent(rows_on_device(imgs, self.device))
            sum_z = z.sum(0) if sum_z is None else (sum_z + z.sum(0))
            n += z.size(0)
            class_counts += lbls.to(self.device).float().sum(0)
            zs.append(z)
            ys.append(lbls.to(self.device).argmax(dim=1))
        # per-class sums (:263-284); the reference's second pass recomputes the same represent()
        z_class_sum = torch.zeros(K, Dz, device=self.device)
        z_class_count = torch.zeros(K, device=self.device)
        for z, y_idx in zip(zs, ys):
            present = torch.bincount(y_idx, minlength=K)[:K].cpu()          # one sync per batch instead of K
            for k in range(K):
                if int(present[k]) > 0:
                    m = (y_idx == k)
                    z_class_sum[k] += z[m].sum(0)                           # the reference's summation order
                    z_class_count[k] += m.sum()
        if _E.dp.active():
            # every rank saw its own shard of the first batches: the counters are sums over rows
            pack = torch.cat([(sum_z if sum_z is not None else torch.zeros(Dz, device=self.device)).double(),
                              torch.tensor([float(n)], device=self.device, dtype=torch.float64),
                              class_counts.double(), z_class_sum.reshape(-1).double(), z_class_count.double()])
            _E.dp.all_reduce_sum(pack)
            sum_z, n = pack[:Dz].float(), int(round(float(pack[Dz])))
            class_counts = pack[Dz + 1:Dz + 1 + K].float()
            z_class_sum = pack[Dz + 1 + K:Dz + 1 + K + K * Dz].float().reshape(K, Dz)
            z_class_count = pack[Dz + 1 + K + K * Dz:].float()
        if n == 0:
            return
        mean_z = (sum_z / n).clamp(1e-4, 1 - 1e-4)                                  # :256
        priors = class_counts / max(1, class_counts.sum())                          # :257
        priors = (priors + 1e-6) / (priors.sum() + 1e-6 * K)                        # :258
        seen = (z_class_count > 0).unsqueeze(1)
        self.z_class_mean = torch.where(seen, z_class_sum / z_class_count.clamp_min(1.0).unsqueeze(1), mean_z.unsqueeze(0))
        self.z_class_count = z_class_count
        self.joint_rbm.vis_bias.data[:Dz] = torch.log(mean_z) - torch.log1p(-mean_z)   # :291
        self.joint_rbm.vis_bias.data[Dz:Dz + K] = torch.log(priors)                    # :292

    # ---- pretrained image stack (imdbn.py:294-384) -----------------------------------------------
    def load_pretrained_image_idbn(self, path: str) -> bool:
        try:
            with open(path, "rb") as f:
                obj = pickle.load(f)
        except Exception as e:
            print(f"[load_pretrained_image_idbn] error: {e}")
            return False
        if isinstance(obj, dict) and "layers" in obj:
            self.image_idbn.layers = obj["layers"]
        elif hasattr(obj, "layers"):
            self.image_idbn = obj
            if not hasattr(self.image_idbn, "text_flag"):
                self.image_idbn.text_flag = False
            if not hasattr(self.image_idbn, "arch_dir"):
                self.image_idbn.arch_dir = os.path.join("logs-idbn", "loaded")
                os.makedirs(self.image_idbn.arch_dir, exist_ok=True)
        else:
            print("[load_pretrained_image_idbn] unrecognized format")
            return False
        for rbm in self.image_idbn.layers:                                          # :325-333
            rbm.W = nn.Parameter(rbm.W.data.to(self.device), requires_grad=False)
            rbm.hid_bias = nn.Parameter(rbm.hid_bias.data.to(self.device), requires_grad=False)
            rbm.vis_bias = nn.Parameter(rbm.vis_bias.data.to(self.device), requires_grad=False)
            rbm.W_m = torch.zeros_like(rbm.W)
            rbm.hb_m = torch.zeros_like(rbm.hid_bias)
            rbm.vb_m = torch.zeros_like(rbm.vis_bias)
            if not hasattr(rbm, "softmax_groups"):
                rbm.softmax_groups = []
        dz_pre = int(self.image_idbn.layers[-1].num_hidden)
        if dz_pre != getattr(self, "Dz_img", dz_pre):
            print(f"[load_pretrained_image_idbn] rebuilding joint: Dz_img -> {dz_pre}")
            self._build_joint(Dz_img=dz_pre, joint_hidden=self.joint_rbm.num_hidden)
        print(f"[load_pretrained_image_idbn] loaded from {path}")
        return True

    def finetune_image_last_layer(self, epochs: int = 0, lr_scale: float = 0.3, cd_k: Optional[int] = None):
        if epochs <= 0:
            return
        last = self.image_idbn.layers[-1]
        old_lr = float(last.lr)
        last.lr = max(1e-8, old_lr * float(lr_scale))                               # :363
        use_cd = int(cd_k) if cd_k is not None else int(self.image_idbn.cd_k)
        print(f"[finetune_image_last_layer] epochs={epochs}, lr={last.lr:.4g}, CD={use_cd}")
        self.finetune_losses = []
        for ep in range(int(epochs)):
            losses = []
            for img, _ in batches(self.dataloader):
                v = rows_on_device(img, self.device)
                for rbm in self.image_idbn.layers[:-1]:
                    v = rbm.forward(v)
                losses.append(last.train_epoch(v, ep, epochs, CD=use_cd))
            if losses:
                self.finetune_losses.append(torch.stack(losses).cpu())
        last.lr = old_lr
        print("[finetune_image_last_layer] done")

    # ---- cross-modal inference (imdbn.py:386-488) -------------------------------------------------
    @torch.no_grad()
    def _cross_reconstruct(self, z_img: torch.Tensor, y_onehot: torch.Tensor,
                           steps: Optional[int] = None, _rbm: Optional[RBM] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Returns (img_from_txt [B,D], p_y_given_img [B,K]).

        The reference's best-of-K refinement is inert (RBM has no ``free_energy``; every candidate
        energy is 0 and argmin picks the main chain, imdbn.py:455-474), so the 4 one-step
        refinement passes are not computed here -- but their draws ARE consumed so that the random
        stream stays aligned with the reference.
        """
        if steps is None:
            steps = self.cross_steps
        B, Dz, K = z_img.size(0), self.Dz_img, self.num_labels
        V = Dz + K
        jr = self.joint_rbm if _rbm is None else _rbm       # (_rbm: train_joint's snapshot of the joint RBM, see _metrics_overlap)
        # IMG -> TXT (:419-427) and TXT -> IMG with mu-pull (:430-449).  The reference runs the two chains one after the other; they
        # share nothing but the joint RBM's read-only weights, so they go to the engine as one call (RBM._chain_pair: side by side
        # in one launch, same draws in the same order, same results)
        v_known = torch.zeros(B, V, device=self.device)
        km = torch.zeros_like(v_known)
        v_known[:, :Dz] = z_img
        km[:, :Dz] = 1.0
        vk_y = torch.zeros(B, V, device=self.device)
        km_y = torch.zeros_like(vk_y)
        vk_y[:, Dz:] = y_onehot
        km_y[:, Dz:] = 1.0
        if getattr(self, "z_class_mean", None) is not None:
            jr._mu_pull = {"mu_k": self.z_class_mean[y_onehot.argmax(dim=1)], "eta0": 0.15}
        else:
            jr._mu_pull = None
        gibbs = dict(v_known=v_known, known_mask=km, n_steps=steps, sample_h=False, sample_v=False)
        nmf = dict(v_known=vk_y, known_mask=km_y, n_steps=steps, T0=3.0, T1=1.0, sigma0=0.9, hot_frac=0.7, sharpen_last=3, T_cold_plus=0.9)
        if hasattr(jr, "_chain_pair"):
            v_img2txt, v_chain = jr._chain_pair(gibbs, nmf)
        else:
            v_img2txt = jr.conditional_gibbs(**gibbs)
            v_chain = jr.noisy_meanfield_annealed(**nmf)
        p_y_given_img = v_img2txt[:, Dz:]
        km = km_y
        if getattr(self, "live_best_of_k", False):
            v_chain = self._best_of_k(v_chain, km, max(1, int(getattr(self, "best_of_k", KBUF))), jr)[0]
        else:
            # dead refinement passes (:460-474): each would draw one U[B,V] at its chain init (rbm.py:333)
            _E.get_engine(jr.W.data).skip_draws(_E.get_rng(), [("u", V)] * (KBUF - 1), B)
        jr._mu_pull = None                                                          # :476
        z_from_y = v_chain[:, :Dz]
        if hasattr(self, "z_affine_scale") and hasattr(self, "z_affine_bias"):       # :481-484
            z_from_y = (z_from_y - self.z_affine_bias) / (self.z_affine_scale + 1e-6)
        return self.image_idbn.decode(z_from_y), p_y_given_img

    @torch.no_grad()
    def _best_of_k(self, v_chain: torch.Tensor, km: torch.Tensor, K: int, _rbm: Optional[RBM] = None):
        """Live version of imdbn.py:451-474: K-1 one-step refinements (T=0.9, no noise), free energy of every
        candidate, per-row argmin, device-side gather (the reference gathers with a Python loop over rows).
        Returns (v_pick [B,V], candidates [K,B,V], energies [K,B])."""
        jr = self.joint_rbm if _rbm is None else _rbm
        cands, energies = [v_chain], [jr.free_energy(v_chain)]
        for _ in range(K - 1):
            v_last = jr.noisy_meanfield_annealed(v_known=cands[-1], known_mask=km, n_steps=1, T0=0.9, T1=0.9,
                                                 sigma0=0.0, hot_frac=0.0, sharpen_last=0, T_cold_plus=0.9)
            cands.append(v_last)
            energies.append(jr.free_energy(v_last))
        cands = torch.stack(cands, dim=0)
        energies = torch.stack(energies, dim=0)
        best = energies.argmin(dim=0)                                     # first minimum, as torch.argmin in the reference
        v_pick = cands[best, torch.arange(cands.size(1), device=cands.device)]
        return v_pick, cands, energies

    @torch.no_grad()
    def represent(self, batch: Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
        """imdbn.py:501-506."""
        img_data, lbl_data = batch
        img = rows_on_device(img_data, self.device)
        y = lbl_data.to(self.device).float()
        return self.joint_rbm.forward(torch.cat([self.image_idbn.represent(img), y], dim=1))

    # ---- joint training (imdbn.py:508-639) --------------------------------------------------------
    def _clamp_y(self, y, B, V, Dz):
        vk = torch.zeros(B, V, device=self.device)
        km = torch.zeros(B, V, device=self.device)
        vk[:, Dz:] = y
        km[:, Dz:] = 1.0
        return vk, km

    def train_joint(self, epochs: int, log_every_pca: int = 25, log_every_probe: int = 10, log_every: int = 5,
                    w_rec: float = 1.0, w_sup: float = 0.0):
        """Warm-up (epochs < 8): 2x label-clamped CD per batch; then free CD + label-clamped CD
        (+ image-clamped CD every 50th batch); `_cross_reconstruct` metrics on EVERY batch.

        Metric accumulators stay on the device and are fetched once per epoch
        (``self.joint_history``); the reference syncs 4x per batch (imdbn.py:635-638).
        """
        print("[iMDBN] joint training (with warmup y-clamp)")
        self.init_joint_bias_from_data(n_batches=10)
        jr = self.joint_rbm
        self.joint_history = []
        # The per-batch metrics (:615-639) only READ the model: `_cross_reconstruct` -- two 50-step chains and a decode, more device time
        # than the batch's updates -- runs on a second stream against a snapshot of the joint RBM (0.55 MB) while the main stream
        # goes on with the next batch's updates, which occupy a fraction of the chip.  Same calls, same draws in the same order, same
        # numbers; one event each way per batch.  JOINT_METRICS_OVERLAP=False runs it in line.
        ov = _MetricsOverlap(self) if (bool(self.params.get("JOINT_METRICS_OVERLAP", True)) and jr.W.is_cuda) else None
        for epoch in range(int(epochs)):
            cd_losses = []
            acc = torch.zeros(5, device=self.device, dtype=torch.float64)   # n, top1, top3, ce_sum, mse_sum
            npix, n_rows = None, 0
            for b_idx, (img, y) in enumerate(batches(self.dataloader)):
                img = rows_on_device(img, self.device)
                y = y.to(self.device).float()
                with torch.no_grad():
                    z_img = self.image_idbn.represent(img)
                    v_plus = torch.cat([z_img, y], dim=1)
                B, Dz, K = z_img.size(0), self.Dz_img, self.num_labels
                V = Dz + K
                aux_cond_steps = int(self.params.get("JOINT_AUX_COND_STEPS", 10))   # :564
                if epoch < WARMUP_Y_EPOCHS:                                          # :566-579
                    for _ in range(2):
                        vk, km = self._clamp_y(y, B, V, Dz)
                        jr.train_epoch_clamped(vk, km, epoch, epochs, CD=1, cond_init_steps=aux_cond_steps,
                                               sample_h=False, sample_v=False, aux_lr_mult=0.3, use_noisy_init=True)
                else:                                                                # :582-612
                    cd_losses.append(jr.train_epoch(v_plus, epoch, epochs, CD=self.joint_cd))
                    vk, km = self._clamp_y(y, B, V, Dz)
                    jr.train_epoch_clamped(vk, km, epoch, epochs, CD=1, cond_init_steps=aux_cond_steps,
                                           sample_h=False, sample_v=False, reclamp_negative=False,
                                           aux_lr_mult=0.3, use_noisy_init=True)
                    if (b_idx % Z_CLAMP_EVERY) == 0:
                        vk.zero_()
                        km.zero_()
                        vk[:, :Dz] = z_img
                        km[:, :Dz] = 1.0
                        jr.train_epoch_clamped(vk, km, epoch, epochs, CD=1, cond_init_steps=aux_cond_steps,
                                               sample_h=False, sample_v=False, reclamp_negative=False,
                                               aux_lr_mult=0.3, use_noisy_init=True)
                npix = img.size(1)
                n_rows += B
                if ov is not None:
                    ov.submit(acc, z_img, y, img)
                else:
                    self._batch_metrics(acc, self.joint_rbm, z_img, y, img)
            if ov is not None:
                ov.join()
            acc[0] = float(n_rows)
            if _E.dp.active():                      # each rank accumulated its shard: sums over rows
                _E.dp.all_reduce_sum(acc)
            a = acc.cpu()
            n = max(1.0, float(a[0]))
            rec = {"epoch": epoch, "n": int(a[0]), "text_top1": float(a[1]) / n, "text_top3": float(a[2]) / n,
                   "text_ce": float(a[3]) / n, "image_mse": float(a[4]) / max(1.0, n * max(1, npix or 1)),
                   "cd_loss": float(torch.stack(cd_losses).mean()) if cd_losses else None,
                   "cd_losses": torch.stack(cd_losses).cpu() if cd_losses else None}
            self.joint_history.append(rec)
            if self.wandb_run:
                if rec["cd_loss"] is not None:
                    self.wandb_run.log({"joint/cd_loss": rec["cd_loss"], "epoch": epoch})
                self.wandb_run.log({"cross_modality/text_top1": rec["text_top1"], "cross_modality/text_top3": rec["text_top3"],
                                    "cross_modality/text_ce": rec["text_ce"], "cross_modality/image_mse": rec["image_mse"],
                                    "epoch": epoch})
        print("[iMDBN] joint training finished.")

    @torch.no_grad()
    def _batch_metrics(self, acc, jr, z_img, y, img):
        """imdbn.py:615-639: the online cross-modal metrics of one batch, accumulated on the device."""
        img_from_txt, p_y = self._cross_reconstruct(z_img, y, steps=self.cross_steps, _rbm=jr)
        gt = y.argmax(dim=1)
        pred = p_y.argmax(dim=1)
        topk_idx = p_y.topk(k=min(3, p_y.size(1)), dim=1).indices
        ce = F.binary_cross_entropy(p_y.clamp(1e-6, 1 - 1e-6),
                                    F.one_hot(gt, num_classes=p_y.size(1)).float(), reduction="sum")
        mse = F.mse_loss(img_from_txt.view_as(img), img, reduction="sum")
        # (the row count is added on the host at the end of the epoch: a `torch.tensor(B, device=...)` per batch is a blocking
        #  host-to-device copy -- it alone cost 1.9 ms per batch and serialised the two streams)
        acc[1:] += torch.stack([(pred == gt).sum().double(),
                                (topk_idx == gt.unsqueeze(1)).any(dim=1).sum().double(),
                                ce.double(), mse.double()])

    # ---- persistence (imdbn.py:815-934) -------------------------------------
    def save_model(self, path: str):
        all_layers = list(self.image_idbn.layers) + [self.joint_rbm]
        payload = {
            "layers": all_layers, "params": self.params,
            "image_idbn": self.image_idbn, "joint_rbm": self.joint_rbm, "num_labels": self.num_labels,
            "Dz_img": self.Dz_img, "arch_str": self.arch_str, "features": self.features,
            "metadata": {"saved_at": datetime.datetime.now().isoformat(), "model_type": "iMDBN",
                         "architecture": self.arch_str},
        }
        for k in ("z_class_mean", "z_affine_scale", "z_affine_bias", "class_names"):
            if getattr(self, k, None) is not None:
                payload[k] = getattr(self, k)
        with open(path, "wb") as f:
            pickle.dump(payload, f)
        print(f"[iMDBN] Model saved to {path}")

    @staticmethod
    def load_model(path: str, device=None) -> Dict[str, Any]:
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        with open(path, "rb") as f:
            payload = pickle.load(f)
        if "image_idbn" in payload:
            for rbm in payload["image_idbn"].layers:
                rbm.to(device)
        if "joint_rbm" in payload:
            payload["joint_rbm"].to(device)
        print(f"[iMDBN] Model loaded from {path}")
        return payload
