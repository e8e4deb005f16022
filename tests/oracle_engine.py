"""Test double for ``imdbn.engine.HipEngine`` built on the CPU oracle.

TEST INFRASTRUCTURE ONLY (lives under tests/, never importable from the product).  It lets the
CPU-only suite run the product's *host* logic -- RBM/iDBN/iMDBN method bodies, step schedules,
draw order, data-parallel packing, pickling -- against the reference fixtures without a GPU.
The generic ``chain``/``clamped_step`` executors below restate the C ABI's step semantics
(include/imdbn_engine.h: imdbn_chain_step) on top of oracle primitives, so a fixture match here
also validates that the step abstraction reproduces rbm.py:240-483.
"""
from __future__ import annotations

import numpy as np
import torch

import oracle.rbm_oracle as O
from oracle.draws import PhiloxStream
from imdbn.engine.rng import PhiloxRng, ReplayRng

F32 = np.float32


def _np(t):
    return t.detach().cpu().numpy().astype(F32, copy=False)


class _Src:
    """uniform / normal / categorical source bound to the engine-level rng object."""

    def __init__(self, rng):
        self.rng = rng
        if isinstance(rng, ReplayRng):
            self.p = rng.provider
        elif isinstance(rng, PhiloxRng):
            self.p = PhiloxStream(rng.seed, rng.offset, rng.row0)
        else:
            raise TypeError(rng)

    def uniform(self, shape): return self.p.uniform(shape)
    def normal(self, shape): return self.p.normal(shape)
    def categorical(self, probs): return self.p.categorical(probs)

    def done(self):
        if isinstance(self.rng, PhiloxRng):
            self.rng.offset = self.p.offset


class OracleEngine:
    name = "oracle-test-double"

    def __init__(self):
        self.calls = []

    # state view sharing memory with the torch tensors (in-place updates propagate)
    @staticmethod
    def _state(rbm, need_m=False) -> O.RBMState:
        W = rbm.W.data
        if need_m:
            for nm, ref in (("W_m", rbm.W.data), ("hb_m", rbm.hid_bias.data), ("vb_m", rbm.vis_bias.data)):
                m = getattr(rbm, nm, None)
                if m is None or m.shape != ref.shape or m.device != ref.device:
                    setattr(rbm, nm, torch.zeros_like(ref))
        z = lambda t: t.numpy()
        return O.RBMState(
            W=z(W), hid_bias=z(rbm.hid_bias.data), vis_bias=z(rbm.vis_bias.data),
            W_m=z(rbm.W_m) if need_m else None, hb_m=z(rbm.hb_m) if need_m else None, vb_m=z(rbm.vb_m) if need_m else None,
            lr=rbm.lr, weight_decay=rbm.weight_decay, momentum=rbm.momentum, dynamic_lr=rbm.dynamic_lr,
            final_momentum=rbm.final_momentum, sparsity=getattr(rbm, "sparsity", False),
            sparsity_factor=getattr(rbm, "sparsity_factor", 0.0),
            softmax_groups=[(int(s), int(e)) for s, e in (getattr(rbm, "softmax_groups", None) or [])])

    @staticmethod
    def _t(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=F32))

    def skip_draws(self, rng, schedule, B):
        s = _Src(rng)
        for kind, n in schedule:
            (s.uniform if kind == "u" else s.normal)((B, n))
        s.done()

    def prop_up(self, rbm, v, T=1.0, sample=False, rng=None):
        st = self._state(rbm)
        p = O.forward(st, _np(v), T)
        if sample:
            s = _Src(rng)
            h = (p > s.uniform(p.shape)).astype(F32)
            s.done()
            return self._t(p), self._t(h)
        return self._t(p)

    def free_energy(self, rbm, v):
        return self._t(O.free_energy(self._state(rbm), _np(v)))

    def prop_down(self, rbm, h, T=1.0, logits_only=False):
        st = self._state(rbm)
        return self._t(O.visible_logits(st, _np(h), T) if logits_only else O.visible_probs(st, _np(h), T))

    def sample_visible(self, rbm, v_prob, rng):
        s = _Src(rng)
        out = O.sample_visible(self._state(rbm), _np(v_prob), s)
        s.done()
        return self._t(out)

    def gibbs_step(self, rbm, v, sample_h, sample_v, rng):
        s = _Src(rng)
        out = O.gibbs_step(self._state(rbm), _np(v), s, sample_h, sample_v)
        s.done()
        return tuple(self._t(x) for x in out)

    def cd_step(self, rbm, data, lr, mom, cd_k, rng, next_data=None):       # the prefetch hint changes no result
        st = self._state(rbm, True)
        s = _Src(rng)
        stats = O.cd_statistics(st, _np(data), cd_k, s)
        s.done()
        O.apply_cd_update(st, stats, lr, mom, stats["n"], st.sparsity)
        return self._t(np.array(stats["sq_err"].mean(dtype=F32))).reshape(())

    # data-parallel split: same packed layout as include/imdbn_engine.h
    def packed_floats(self, V, H):
        return ((V * H + 2 * H + V + 1) + 3) // 4 * 4

    def cd_stats(self, rbm, data, cd_k, rng, out=None):
        st = self._state(rbm)
        s = _Src(rng)
        x = O.cd_statistics(st, _np(data), cd_k, s)
        s.done()
        V, H = st.W.shape
        p = np.zeros(self.packed_floats(V, H), F32)
        p[:V * H] = (x["pos_assoc"] - x["neg_assoc"]).ravel()
        o = V * H
        p[o:o + H] = x["pos_h_sum"] - x["neg_h_sum"]
        p[o + H:o + H + V] = x["data_sum"] - x["v_sum"]
        p[o + H + V:o + 2 * H + V] = x["pos_h_sum"]
        p[o + 2 * H + V] = x["sq_err"].sum(dtype=F32)
        return self._t(p)

    # ---- factor exchange (host-logic stand-in: a rank's "factor block" is its packed statistics as bytes) ----
    def factor_mode_ok(self, rbm, B) -> bool:
        return 1 <= B <= 64 and not (getattr(rbm, "softmax_groups", None) or [])

    def cd_factors(self, rbm, data, cd_k, rng):
        return self.cd_stats(rbm, data, cd_k, rng).contiguous().view(torch.uint8)

    def gather_buffer(self, rbm, B, world):
        V, H = rbm.W.shape
        return torch.empty(world, 4 * self.packed_floats(V, H), dtype=torch.uint8)

    def apply_factors(self, rbm, gathered, rows_per_rank, global_B, lr, mom):
        assert gathered.dtype == torch.uint8 and gathered.dim() == 2
        packed = gathered.view(torch.float32).sum(0)
        return self.apply_delta(rbm, packed, global_B, lr, mom)

    def apply_delta(self, rbm, packed, global_B, lr, mom, sparsity=None):
        st = self._state(rbm, True)
        use_sparsity = st.sparsity if sparsity is None else sparsity
        V, H = st.W.shape
        p = _np(packed)
        o = V * H
        n = F32(global_B)
        lr32, mom32 = F32(lr), F32(mom)
        st.W_m *= mom32
        st.W_m += lr32 * (p[:o].reshape(V, H) / n - F32(st.weight_decay) * st.W)
        st.W += st.W_m
        st.hb_m *= mom32
        st.hb_m += lr32 * p[o:o + H] / n
        if use_sparsity:
            st.hb_m += F32(-lr) * (p[o + H + V:o + 2 * H + V] / n - F32(st.sparsity_factor))
        st.hid_bias += st.hb_m
        st.vb_m *= mom32
        st.vb_m += lr32 * p[o + H:o + H + V] / n
        st.vis_bias += st.vb_m
        return self._t(np.array(p[o + 2 * H + V] / (n * F32(V)), F32)).reshape(())

    # ---- generic step executor (imdbn_chain_step semantics) ------------------------------------
    @staticmethod
    def _step(st, v, vk, km, s, step, mu):
        T = max(1e-6, step["T"])
        hl = ((v @ st.W) + st.hid_bias) / F32(T)
        if step["sigma"] > 0:
            hl = hl + s.normal(hl.shape) * F32(step["sigma"])
        p_h = O.sigmoid(hl)
        h = (p_h > s.uniform(p_h.shape)).astype(F32) if step["sample_h"] else p_h
        vl = ((h @ st.W.T) + st.vis_bias) / F32(T)
        if step["sigma"] > 0:
            vl = vl + s.normal(vl.shape) * F32(step["sigma"])
        vl = vl.astype(F32)
        p_v = O._apply_groups(st, O.sigmoid(vl), vl)
        if mu is not None and step["eta"] != 0.0:
            Dz = mu.shape[1]
            p_v[:, :Dz] = F32(1 - step["eta"]) * p_v[:, :Dz] + F32(step["eta"]) * mu
        mix = (lambda x: (x * (F32(1) - km) + vk * km).astype(F32)) if step["clamp"] else (lambda x: x)
        if step["vmode"] == 0:
            out = mix(p_v)
        elif step["vmode"] == 1:
            out = mix(O.sample_visible(st, p_v, s))
        else:
            out = O.sample_visible(st, mix(p_v), s)
        return out, p_v, h, p_h

    def _run_chain(self, st, vk, km, steps, s, init_uniform, mu):
        v = (vk * km + (F32(1) - km) * s.uniform(vk.shape)).astype(F32) if init_uniform else vk.copy()
        for step in steps:
            v = self._step(st, v, vk, km, s, step, mu)[0]
        return v

    def chain(self, rbm, v_known, mask, steps, rng, init_uniform=True, mu=None):
        st = self._state(rbm)
        s = _Src(rng)
        v = self._run_chain(st, _np(v_known), _np(mask), steps, s, init_uniform, None if mu is None else _np(mu))
        s.done()
        return self._t(v)

    def _clamped_stats(self, st, v_known, mask, init_steps, mu, cd_k, sample_h, sample_v, reclamp, rng):
        s = _Src(rng)
        vk, km = _np(v_known), _np(mask)
        v_plus = self._run_chain(st, vk, km, init_steps, s, True, None if mu is None else _np(mu))
        h_plus = O.forward(st, v_plus)
        v_neg = v_plus
        for _ in range(int(cd_k)):
            step = dict(T=1.0, sigma=0.0, eta=0.0, sample_h=sample_h, vmode=2 if sample_v else 0, clamp=reclamp)
            v_neg = self._step(st, v_neg, vk, km, s, step, None)[0]
        h_neg = O.forward(st, v_neg)
        s.done()
        stats = dict(pos_assoc=(v_plus.T @ h_plus).astype(F32), neg_assoc=(v_neg.T @ h_neg).astype(F32),
                     pos_h_sum=h_plus.sum(0, dtype=F32), neg_h_sum=h_neg.sum(0, dtype=F32),
                     data_sum=v_plus.sum(0, dtype=F32), v_sum=v_neg.sum(0, dtype=F32))
        return stats, v_plus, v_neg

    def clamped_step(self, rbm, v_known, mask, init_steps, mu, lr, mom, cd_k, sample_h, sample_v, reclamp, rng):
        st = self._state(rbm, True)
        stats, v_plus, v_neg = self._clamped_stats(st, v_known, mask, init_steps, mu, cd_k, sample_h, sample_v, reclamp, rng)
        O.apply_cd_update(st, stats, lr, mom, v_plus.shape[0], False)
        return self._t(np.array(((v_plus - v_neg) ** 2).mean(dtype=F32), F32)).reshape(())

    def clamped_stats(self, rbm, v_known, mask, init_steps, mu, cd_k, sample_h, sample_v, reclamp, rng, out=None):
        st = self._state(rbm)
        x, v_plus, v_neg = self._clamped_stats(st, v_known, mask, init_steps, mu, cd_k, sample_h, sample_v, reclamp, rng)
        V, H = st.W.shape
        p = np.zeros(self.packed_floats(V, H), F32)
        o = V * H
        p[:o] = (x["pos_assoc"] - x["neg_assoc"]).ravel()
        p[o:o + H] = x["pos_h_sum"] - x["neg_h_sum"]
        p[o + H:o + H + V] = x["data_sum"] - x["v_sum"]
        p[o + H + V:o + 2 * H + V] = x["pos_h_sum"]
        p[o + 2 * H + V] = ((v_plus - v_neg) ** 2).sum(dtype=F32)
        return self._t(p)
