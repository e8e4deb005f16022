"""RBM with the reference's class surface, backed by the MI355X CD engine.

API mirror of the reference ``imdbn/models/rbm.py`` (same constructor, method names, keyword
defaults, attribute names and pickle state -- SURVEY.md 8b-1 / Appendix C), but every method body
is host-side scalar logic plus ONE call into the native engine (``imdbn.engine``): no ATen
arithmetic on the hot path.  Reference line numbers are cited per method.

Deliberate differences: random draws come from the engine's RNG source
(``imdbn.engine.get_rng()``: device Philox by default, recorded draws in the parity tests) instead
of torch's global generator; all returned tensors are grad-less.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from imdbn import engine as _E


def _row_pitch(h: int) -> int:
    """Row pitch in elements: H rounded up to 128 floats (512 B: the row segment one half-wave of the update kernel
    reads and writes) or else to 32 floats (128 B), whichever costs < 7 % extra memory."""
    forced = int(os.environ.get("IMDBN_ROW_PITCH_ABS", "0"))       # layout experiments: an absolute pitch (>= h, multiple of 4)
    if forced >= h and forced % 4 == 0:
        return forced
    for q in (int(os.environ.get("IMDBN_ROW_PITCH", "128")), 32):
        p = (h + q - 1) // q * q
        if (p - h) * 16 <= h:
            return p
    return h


def _step(T=1.0, sigma=0.0, eta=0.0, sample_h=False, vmode=0, clamp=True) -> dict:
    return {"T": float(T), "sigma": float(sigma), "eta": float(eta), "sample_h": bool(sample_h),
            "vmode": int(vmode), "clamp": bool(clamp)}


class RBM(nn.Module):
    """Bernoulli RBM with optional softmax groups on the visible layer (reference rbm.py:24-79)."""

    def __init__(
        self,
        num_visible: int,
        num_hidden: int,
        learning_rate: float,
        weight_decay: float,
        momentum: float,
        dynamic_lr: bool = False,
        final_momentum: float = 0.97,
        sparsity: bool = False,
        sparsity_factor: float = 0.05,
        softmax_groups: Optional[List[Tuple[int, int]]] = None,
    ):
        super().__init__()
        self.num_visible = int(num_visible)
        self.num_hidden = int(num_hidden)
        self.lr = float(learning_rate)
        self.weight_decay = float(weight_decay)
        self.momentum = float(momentum)
        self.dynamic_lr = bool(dynamic_lr)
        self.final_momentum = float(final_momentum)
        self.sparsity = bool(sparsity)
        self.sparsity_factor = float(sparsity_factor)
        self.softmax_groups = softmax_groups or []

        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")      # rbm.py:69
        # Same law as the reference (randn / sqrt(V), rbm.py:70-72).  On the GPU the rows of W and W_m are
        # allocated with a pitch padded to 128 B: shape, values and semantics are unchanged (W is a [V,H]
        # view with stride (pitch, 1)), but every 512-B row segment the kernels touch is cache-line aligned
        # (H = 1500 would otherwise straddle 5 lines instead of 4: ~25 % over-fetch, profiles/r01_pmc_*).
        pitch = _row_pitch(self.num_hidden) if device.type == "cuda" else self.num_hidden
        w_full = torch.randn(self.num_visible, pitch, device=device) / math.sqrt(max(1, self.num_visible))
        self.W = nn.Parameter(w_full[:, :self.num_hidden], requires_grad=False)
        self.hid_bias = nn.Parameter(torch.zeros(self.num_hidden, device=device), requires_grad=False)
        self.vis_bias = nn.Parameter(torch.zeros(self.num_visible, device=device), requires_grad=False)
        # momentum buffers: plain attributes, exactly as in the reference (rbm.py:77-79)
        self.W_m = torch.zeros(self.num_visible, pitch, device=device)[:, :self.num_hidden]
        self.hb_m = torch.zeros_like(self.hid_bias)
        self.vb_m = torch.zeros_like(self.vis_bias)

    # ---- pickle state ----------------------------------------------------
    def __getstate__(self):
        """The reference's plain attribute state.  On the GPU ``W`` / ``W_m`` are views of row-padded buffers; a pickle
        must not carry that storage: they are written as contiguous ``[V, H]`` tensors (stride ``(H, 1)``), exactly what
        the reference class -- or this one -- expects to unpickle."""
        state = self.__dict__.copy()
        state.pop("_imdbn_desc", None)                 # the engine's cached native descriptor (raw addresses): never part of the state
        params = state["_parameters"].copy()
        W = params.get("W")
        if W is not None and not W.is_contiguous():
            params["W"] = nn.Parameter(W.data.contiguous(), requires_grad=W.requires_grad)
        state["_parameters"] = params
        Wm = state.get("W_m")
        if isinstance(Wm, torch.Tensor) and not Wm.is_contiguous():
            state["W_m"] = Wm.contiguous()
        return state

    # ---- helpers ------------------------------------------------------------------------------
    def _eng(self):
        return _E.get_engine(self.W.data)

    def _in(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(device=self.W.device, dtype=torch.float32)

    def _mu(self):
        mp = getattr(self, "_mu_pull", None)                                        # rbm.py:359
        if mp is None:
            return None, 0.0
        return self._in(mp["mu_k"]), float(mp.get("eta0", 0.15))

    def _lr_mom(self, epoch: int):
        lr = self.lr / (1 + 0.01 * epoch) if self.dynamic_lr else self.lr          # rbm.py:194
        mom = self.momentum if epoch <= 5 else self.final_momentum                  # rbm.py:195
        return lr, mom

    # ---- propagations (rbm.py:81-178) ---------------------------------------------------------
    @torch.no_grad()
    def forward(self, v: torch.Tensor, T: float = 1.0) -> torch.Tensor:
        """p(h|v) = sigmoid((v W + c)/max(1e-6,T))   (rbm.py:92)."""
        eng = self._eng()
        if T == 1.0 and hasattr(eng, "forward") and hasattr(eng, "binary_hint"):
            # the same path as the forward fused into train_epoch(return_forward=True): bit-identical results
            out = eng.forward(self, self._in(v), data_binary=eng.binary_hint(v))
        else:
            out = eng.prop_up(self, self._in(v), T=T)
        out._imdbn_binary = False      # probabilities: the next layer's update makes no bit plane of them (HipEngine.binary_hint)
        return out

    @torch.no_grad()
    def _visible_logits(self, h: torch.Tensor, T: float = 1.0) -> torch.Tensor:
        """(h W^T + b)/max(1e-6,T)   (rbm.py:96)."""
        return self._eng().prop_down(self, self._in(h), T=T, logits_only=True)

    @torch.no_grad()
    def visible_probs(self, h: torch.Tensor, T: float = 1.0) -> torch.Tensor:
        """p(v|h) with softmax over each group (rbm.py:109-116)."""
        return self._eng().prop_down(self, self._in(h), T=T)

    @torch.no_grad()
    def free_energy(self, v: torch.Tensor) -> torch.Tensor:
        """F(v) = -v.b - sum_j softplus(c_j + (vW)_j), shape [B] (imdbn/utils/energy_utils.py:19-28).

        The reference's ``RBM`` has no such method (``iMDBN._cross_reconstruct`` probes for it with ``hasattr``,
        imdbn.py:455, and falls back to zeros); here it exists and feeds the opt-in live best-of-K selection
        (``iMDBN.live_best_of_k``)."""
        return self._eng().free_energy(self, self._in(v))

    @torch.no_grad()
    def sample_visible(self, v_prob: torch.Tensor) -> torch.Tensor:
        """Bernoulli over all columns, one categorical per softmax group (rbm.py:125-135)."""
        return self._eng().sample_visible(self, self._in(v_prob), self._rng(v_prob.size(0)))

    @torch.no_grad()
    def backward(self, h: torch.Tensor, return_logits: bool = False) -> torch.Tensor:
        """Decoder alias of visible_probs at T=1 (rbm.py:148-151)."""
        if return_logits:
            return self._visible_logits(h)
        return self.visible_probs(h)

    @torch.no_grad()
    def backward_sample(self, h: torch.Tensor) -> torch.Tensor:
        """rbm.py:156."""
        return self.sample_visible(self.visible_probs(h))

    @torch.no_grad()
    def gibbs_step(self, v: torch.Tensor, sample_h: bool = True, sample_v: bool = True):
        """One v -> h -> v' step; returns (v_next, v_prob, h, h_prob)   (rbm.py:174-178)."""
        return self._eng().gibbs_step(self, self._in(v), sample_h, sample_v, self._rng(v.size(0)))

    def _rng(self, B: int):
        """The ambient draw source; under data parallelism Philox draws are keyed on the GLOBAL batch row
        (rank r owns rows [r*B, (r+1)*B)), so a sharded call equals the unsharded one."""
        rng = _E.get_rng()
        if _E.dp.active() and isinstance(rng, _E.PhiloxRng):
            rng.row0 = _E.dp.rank() * int(B)
        return rng

    # ---- CD-k update (rbm.py:180-227) -----------------------------------------------------------
    @torch.no_grad()
    def train_epoch(self, data: torch.Tensor, epoch: int, max_epochs: int, CD: int = 1,
                    next_data: Optional[torch.Tensor] = None, return_forward: bool = False):
        """One CD-k update on one mini-batch (the name is the reference's); returns the 0-d MSE loss.

        ``next_data`` (extension, optional): the batch the following ``train_epoch`` call will receive.  Its operand
        forms are prepared by extra blocks of one of this call's launches; hand that same tensor, unmodified, to the next
        call and it skips its own preparation (results are bit-identical either way; a modified, different or
        ineligible tensor is simply prepared again).

        ``return_forward`` (extension): return ``(loss, self.forward(data))`` -- the pair of calls of the layer loop
        (idbn.py:195-204) -- with the forward pass run inside the same engine call where it can be (the batch's operand
        forms are still in the workspace); bit-identical to the two separate calls.

        With data parallelism enabled (``imdbn.engine.dp.enable()``) ``data`` is this rank's shard
        of the global batch: the ranks exchange their factor blocks (or all-reduce the statistics, see
        ``imdbn.engine.dp.enable``) once and every replica applies the same update with 1/global_batch
        (SURVEY.md 8e).
        """
        lr, mom = self._lr_mom(epoch)
        eng, x = self._eng(), self._in(data)
        rng = self._rng(x.size(0))
        # 0/1 batches (binary images) are read as bit planes by the positive phase; what a batch contains is found out on the
        # device (imdbn_cd_opts.data_binary = unknown) unless the caller's tensor object carries a loader's tag (``_in`` makes a
        # fresh view every call)
        kw = {"data_binary": eng.binary_hint(data)} if hasattr(eng, "binary_hint") else {}
        dp = _E.dp
        if dp.active():
            B = x.size(0)
            dp.validate_rows(B, x.device)
            ret = (lambda loss: (loss, self.forward(data))) if return_forward else (lambda loss: loss)
            if dp.mode() == "factors" and hasattr(eng, "factor_mode_ok") and eng.factor_mode_ok(self, B):
                # exchange the factors (~7 MB per rank at 10000 x 1500; 2 MB in wire form) instead of the fp32 statistics (60 MB)
                if hasattr(eng, "cd_factors_wire"):
                    # three calls per step: CD pass into the wire form (with the next-batch hint), all-gather, update
                    binary = dp.binary_data()
                    wire = eng.cd_factors_wire(self, x, CD, rng, binary, next_data=next_data, **kw)
                    wires = dp.all_gather_blocks(eng.compact_gather_buffer(self, B, dp.world_size(), binary), wire)
                    return ret(eng.apply_wire(self, wires, B, B * dp.world_size(), binary, lr, mom))
                block = eng.cd_factors(self, x, CD, rng, **kw)
                if hasattr(eng, "pack_factors"):
                    # wire form: the visible planes as bits (the sample always, the data when declared binary)
                    binary = dp.binary_data()
                    wire = eng.pack_factors(self, block, B, binary)
                    wires = dp.all_gather_blocks(eng.compact_gather_buffer(self, B, dp.world_size(), binary), wire)
                    if hasattr(eng, "apply_factors_wire"):       # the head of the blocks is read from the wire blocks in place
                        planes = eng.unpack_factors(self, wires, B, binary, planes_only=True)
                        return ret(eng.apply_factors_wire(self, wires, planes, B, B * dp.world_size(), lr, mom))
                    gathered = eng.unpack_factors(self, wires, B, binary)
                else:
                    gathered = dp.all_gather_blocks(eng.gather_buffer(self, B, dp.world_size()), block)
                return ret(eng.apply_factors(self, gathered, B, B * dp.world_size(), lr, mom))
            buf = eng.packed_buffer(self) if hasattr(eng, "packed_buffer") else None
            packed = eng.cd_stats(self, x, CD, rng, out=buf, **kw, **({"next_data": next_data} if (next_data is not None and hasattr(eng, "prefetch_ok")) else {}))
            dp.all_reduce_sum(packed)
            return ret(eng.apply_delta(self, packed, B * dp.world_size(), lr, mom))
        if return_forward:
            if not getattr(eng, "fused_forward", False):
                loss = eng.cd_step(self, x, lr, mom, CD, rng, **({"next_data": next_data} if next_data is not None else {}), **kw)
                return loss, self.forward(data)
            loss, h = eng.cd_step(self, x, lr, mom, CD, rng, next_data=next_data, forward=True, **kw)
            h._imdbn_binary = False
            return loss, h
        if next_data is not None:
            return eng.cd_step(self, x, lr, mom, CD, rng, next_data=next_data, **kw)
        return eng.cd_step(self, x, lr, mom, CD, rng, **kw)

    # ---- schedules (rbm.py:229-238) -------------------------------------------------------------
    def _lin_schedule(self, t, t_max, start, end):
        if t_max <= 1:
            return float(end)
        alpha = min(max(t / (t_max - 1), 0.0), 1.0)
        return float(start + (end - start) * alpha)

    def _hot_steps(self, n_steps, hot_frac):
        return int(max(0, min(n_steps, round(hot_frac * n_steps))))

    def _nmf_steps(self, n_steps, T0, T1, sigma0, sharpen_last, T_cold_plus, eta0) -> List[dict]:
        """Host scalars of rbm.py:337-341,362 for each step of noisy mean-field annealing."""
        n = int(n_steps)
        steps = []
        for t in range(n):
            Tt = self._lin_schedule(t, n, T0, T1)
            if (n - t) <= max(1, int(sharpen_last)):
                Tt = T_cold_plus
            frac = max(0.0, 1.0 - (t / max(1, n - 1)))
            steps.append(_step(T=Tt, sigma=sigma0 * frac, eta=eta0 * frac, sample_h=False, vmode=0, clamp=True))
        return steps

    # ---- chains (rbm.py:240-400) ---------------------------------------------------------------
    @torch.no_grad()
    def conditional_gibbs_annealed(
        self,
        v_known: torch.Tensor,
        known_mask: torch.Tensor,
        n_steps: int = 40,
        T0: float = 2.5,
        T1: float = 1.0,
        sample_h_until: int = 20,
        sample_v_every: int = 0,
        final_meanfield: bool = True,
    ):
        """rbm.py:270-298."""
        n = int(n_steps)
        hot = int(max(0, min(n_steps, sample_h_until)))
        steps = []
        for t in range(n):
            Tt = self._lin_schedule(t, n, T0, T1)
            if (n - t) <= 3:
                Tt = min(0.9, Tt)
            sv = (t < hot) and (sample_v_every > 0) and (t % sample_v_every == 0)
            steps.append(_step(T=Tt, sample_h=(t < hot), vmode=1 if sv else 0, clamp=True))
        if final_meanfield:
            steps.append(_step(T=1.0, clamp=True))
        return self._eng().chain(self, self._in(v_known), self._in(known_mask), steps, self._rng(v_known.size(0)))

    @torch.no_grad()
    def noisy_meanfield_annealed(
        self,
        v_known: torch.Tensor,
        known_mask: torch.Tensor,
        n_steps: int = 72,
        T0: float = 3.0,
        T1: float = 1.0,
        sigma0: float = 0.9,
        hot_frac: float = 0.7,
        sharpen_last: int = 3,
        T_cold_plus: float = 0.9,
    ):
        """rbm.py:332-367 (``hot_frac`` is accepted and, as in the reference, has no effect)."""
        mu, eta0 = self._mu()
        steps = self._nmf_steps(n_steps, T0, T1, sigma0, sharpen_last, T_cold_plus, eta0 if mu is not None else 0.0)
        return self._eng().chain(self, self._in(v_known), self._in(known_mask), steps, self._rng(v_known.size(0)), mu=mu)

    @torch.no_grad()
    def conditional_gibbs(
        self,
        v_known: torch.Tensor,
        known_mask: torch.Tensor,
        n_steps: int = 30,
        sample_h: bool = False,
        sample_v: bool = False,
    ) -> torch.Tensor:
        """rbm.py:391-400; the last entry is the reference's final UN-clamped pass (:400)."""
        steps = [_step(sample_h=sample_h, vmode=1 if sample_v else 0, clamp=True) for _ in range(int(n_steps))]
        steps.append(_step(clamp=False))
        return self._eng().chain(self, self._in(v_known), self._in(known_mask), steps, self._rng(v_known.size(0)))

    @torch.no_grad()
    def _chain_pair(self, gibbs: dict, nmf: dict):
        """``conditional_gibbs(**gibbs)`` and ``noisy_meanfield_annealed(**nmf)`` (run in that order by the reference,
        imdbn.py:424-449) as ONE engine call where the engine offers it: the two chains are independent (they share only the
        read-only weights), so they run side by side in one launch.  Same draws in the same order, same results as the two calls.
        Returns (v_gibbs, v_nmf)."""
        eng = self._eng()
        if not hasattr(eng, "chain_pair"):
            return self.conditional_gibbs(**gibbs), self.noisy_meanfield_annealed(**nmf)
        g = dict(n_steps=30, sample_h=False, sample_v=False); g.update(gibbs)
        n = dict(n_steps=72, T0=3.0, T1=1.0, sigma0=0.9, hot_frac=0.7, sharpen_last=3, T_cold_plus=0.9); n.update(nmf)
        steps_g = [_step(sample_h=g["sample_h"], vmode=1 if g["sample_v"] else 0, clamp=True) for _ in range(int(g["n_steps"]))]
        steps_g.append(_step(clamp=False))                                          # rbm.py:400
        mu, eta0 = self._mu()
        steps_n = self._nmf_steps(n["n_steps"], n["T0"], n["T1"], n["sigma0"], n["sharpen_last"], n["T_cold_plus"], eta0 if mu is not None else 0.0)
        a = {"v_known": self._in(g["v_known"]), "mask": self._in(g["known_mask"]), "steps": steps_g}
        b = {"v_known": self._in(n["v_known"]), "mask": self._in(n["known_mask"]), "steps": steps_n, "mu": mu}
        return eng.chain_pair(self, a, b, self._rng(a["v_known"].size(0)))

    # ---- clamped CD (rbm.py:402-483) -------------------------------------------------------------
    @torch.no_grad()
    def train_epoch_clamped(
        self,
        v_known: torch.Tensor,
        known_mask: torch.Tensor,
        epoch: int,
        max_epochs: int,
        CD: int = 1,
        cond_init_steps: int = 50,
        sample_h: bool = True,
        sample_v: bool = False,
        reclamp_negative: bool = True,
        aux_lr_mult: float = 0.3,
        use_noisy_init: bool = True,
    ):
        """Auxiliary clamped CD update; returns the 0-d loss mean((v+ - v-)^2)."""
        lr, mom = self._lr_mom(epoch)
        mu = None
        if use_noisy_init:                                                          # rbm.py:443-448
            mu, eta0 = self._mu()
            init = self._nmf_steps(max(10, int(cond_init_steps)), 3.0, 1.0, 0.9, 2, 0.9, eta0 if mu is not None else 0.0)
        else:                                                                       # rbm.py:450-453
            init = [_step(sample_h=sample_h, vmode=1 if sample_v else 0, clamp=True) for _ in range(int(cond_init_steps))]
            init.append(_step(clamp=False))
        eng, vk, km = self._eng(), self._in(v_known), self._in(known_mask)
        rng = self._rng(vk.size(0))
        dp = _E.dp
        if dp.active():
            # this rank's rows of the global batch; one all-reduce of the packed statistics
            B = vk.size(0)
            dp.validate_rows(B, vk.device)
            buf = eng.packed_buffer(self) if hasattr(eng, "packed_buffer") else None
            packed = eng.clamped_stats(self, vk, km, init, mu, CD, sample_h, sample_v, reclamp_negative, rng, out=buf)
            dp.all_reduce_sum(packed)
            return eng.apply_delta(self, packed, B * dp.world_size(), aux_lr_mult * lr, mom, sparsity=False)
        return eng.clamped_step(self, vk, km, init, mu, aux_lr_mult * lr, mom, CD, sample_h, sample_v, reclamp_negative, rng)
