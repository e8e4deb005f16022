"""CPU oracle: a numpy/float32 restatement of the reference's contrastive-divergence path.

TEST INFRASTRUCTURE ONLY -- never imported by the product path
(``multimodal-idbn_amd/``).  Allowed importers: ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` (as the checker / the reported CPU
baseline, never as the thing shipped).

Every function cites the reference lines it follows (paths relative to
``/root/reference/imdbn/models/``).  Arithmetic is kept operation-for-operation in
float32 in the reference's order; random draws come from an
explicit source (``oracle/draws.py``) in the reference's call order (Appendix B).

Parity pin: ``tests/test_oracle_golden.py`` checks this file against golden vectors
produced by importing the *unmodified* reference in the build container
(``tests/golden/make_fixtures.py``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32

# Smallest |p - u| seen at a Bernoulli draw since the last reset: an accelerator result that differs from this
# restatement by ONE flipped sample is legitimate when that margin is at rounding level (tools/stress_parity.py, fixtures).
BERNOULLI_MARGIN = {"min": float("inf")}


def reset_margin():
    BERNOULLI_MARGIN["min"] = float("inf")


# Near ties: 1[p > u] is discontinuous, and at full size (~10^6 comparisons per update) some |p - u| are
# at fp32 rounding level, where the summation order of the propagation decides.  A test that holds the samples the
# accelerator actually drew can queue them here (one array per Bernoulli tensor, in draw order; None = no override):
# wherever |p - u| < tol the queued decision is taken instead of this restatement's own.  `used` counts how many
# decisions that changed, `ties` how many elements were within tol -- the caller asserts that both stay tiny.
TIE_BREAK = {"tol": 0.0, "queue": [], "used": 0, "ties": 0}


def set_tie_break(decisions=None, tol: float = 2e-6):
    TIE_BREAK["tol"] = float(tol)
    TIE_BREAK["queue"] = list(decisions or [])
    TIE_BREAK["used"] = 0
    TIE_BREAK["ties"] = 0


def _bern(p, u):
    """1[p > u] as float32, tracking the smallest margin."""
    if p.size:
        BERNOULLI_MARGIN["min"] = min(BERNOULLI_MARGIN["min"], float(np.abs(p - u).min()))
    out = p > u
    if TIE_BREAK["queue"]:
        d = TIE_BREAK["queue"].pop(0)
        if d is not None:
            tie = np.abs(p - u) < F32(TIE_BREAK["tol"])
            d = np.asarray(d).astype(bool).reshape(out.shape)
            TIE_BREAK["ties"] += int(tie.sum())
            TIE_BREAK["used"] += int((tie & (out != d)).sum())
            out = np.where(tie, d, out)
    return out.astype(F32)



def _f(x) -> np.float32:
    return np.float32(x)


# ---------------------------------------------------------------------------
# state  (rbm.py:41-79)
# ---------------------------------------------------------------------------
@dataclass
class RBMState:
    W: np.ndarray                 # [V, H] float32, row-major
    hid_bias: np.ndarray          # [H]
    vis_bias: np.ndarray          # [V]
    W_m: np.ndarray
    hb_m: np.ndarray
    vb_m: np.ndarray
    lr: float = 0.1
    weight_decay: float = 1e-4
    momentum: float = 0.5
    dynamic_lr: bool = False
    final_momentum: float = 0.97
    sparsity: bool = False
    sparsity_factor: float = 0.05
    softmax_groups: List[Tuple[int, int]] = field(default_factory=list)
    mu_pull: Optional[dict] = None   # {"mu_k": [B,Dz], "eta0": float}  (rbm.py:359-363)

    @property
    def num_visible(self) -> int:
        return self.W.shape[0]

    @property
    def num_hidden(self) -> int:
        return self.W.shape[1]

    @staticmethod
    def create(W, lr, weight_decay, momentum, dynamic_lr=False, final_momentum=0.97,
               sparsity=False, sparsity_factor=0.05, softmax_groups=None,
               hid_bias=None, vis_bias=None) -> "RBMState":
        W = np.ascontiguousarray(W, dtype=F32)
        V, H = W.shape
        return RBMState(
            W=W.copy(),
            hid_bias=np.zeros(H, F32) if hid_bias is None else np.asarray(hid_bias, F32).copy(),
            vis_bias=np.zeros(V, F32) if vis_bias is None else np.asarray(vis_bias, F32).copy(),
            W_m=np.zeros((V, H), F32), hb_m=np.zeros(H, F32), vb_m=np.zeros(V, F32),
            lr=float(lr), weight_decay=float(weight_decay), momentum=float(momentum),
            dynamic_lr=bool(dynamic_lr), final_momentum=float(final_momentum),
            sparsity=bool(sparsity), sparsity_factor=float(sparsity_factor),
            softmax_groups=[tuple(int(x) for x in g) for g in (softmax_groups or [])],
        )

    def copy(self) -> "RBMState":
        c = RBMState(**{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in self.__dict__.items()})
        c.softmax_groups = list(self.softmax_groups)
        return c


# ---------------------------------------------------------------------------
# element-wise pieces
# ---------------------------------------------------------------------------
def sigmoid(x: np.ndarray) -> np.ndarray:
    """rbm.py:19-21 (custom) and torch.sigmoid (rbm.py:110,347,354): 1/(1+exp(-x)) in f32."""
    x = np.asarray(x, F32)
    with np.errstate(over="ignore"):
        return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def _softmax_rows(l: np.ndarray) -> np.ndarray:
    """torch.softmax(dim=1): exp(x - max) / sum (rbm.py:114,356)."""
    m = l.max(axis=1, keepdims=True)
    e = np.exp((l - m).astype(F32)).astype(F32)
    return (e / e.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)


def _temp(T: float) -> np.float32:
    return F32(max(1e-6, T))


# ---------------------------------------------------------------------------
# propagations  (rbm.py:81-116, 137-151)
# ---------------------------------------------------------------------------
def forward(st: RBMState, v: np.ndarray, T: float = 1.0) -> np.ndarray:
    """rbm.py:92  sigmoid((v @ W + hid_bias) / max(1e-6, T))."""
    v = np.asarray(v, F32)
    return sigmoid(((v @ st.W) + st.hid_bias) / _temp(T))


def visible_logits(st: RBMState, h: np.ndarray, T: float = 1.0) -> np.ndarray:
    """rbm.py:96  (h @ W.T + vis_bias) / max(1e-6, T)."""
    h = np.asarray(h, F32)
    return (((h @ st.W.T) + st.vis_bias) / _temp(T)).astype(F32)


def _apply_groups(st: RBMState, probs: np.ndarray, logits: np.ndarray) -> np.ndarray:
    for s, e in st.softmax_groups:
        probs[:, s:e] = _softmax_rows(logits[:, s:e])
    return probs


def visible_probs(st: RBMState, h: np.ndarray, T: float = 1.0) -> np.ndarray:
    """rbm.py:109-116."""
    logits = visible_logits(st, h, T)
    return _apply_groups(st, sigmoid(logits), logits)


def sample_visible(st: RBMState, v_prob: np.ndarray, rng) -> np.ndarray:
    """rbm.py:125-135: Bernoulli over ALL columns, then one categorical per softmax group."""
    u = rng.uniform(v_prob.shape)
    v = _bern(v_prob, u)
    for s, e in st.softmax_groups:
        probs = np.clip(v_prob[:, s:e], F32(1e-8), F32(1.0)).astype(F32)
        idx = rng.categorical(probs)
        v[:, s:e] = 0.0
        v[np.arange(v.shape[0]), s + idx] = 1.0
    return v


def backward(st: RBMState, h: np.ndarray, return_logits: bool = False) -> np.ndarray:
    """rbm.py:148-151."""
    if return_logits:
        return visible_logits(st, h)
    return visible_probs(st, h)


def backward_sample(st: RBMState, h, rng):
    """rbm.py:156."""
    return sample_visible(st, visible_probs(st, h), rng)


def gibbs_step(st: RBMState, v, rng, sample_h=True, sample_v=True):
    """rbm.py:174-178."""
    h_prob = forward(st, v)
    h = _bern(h_prob, rng.uniform(h_prob.shape)) if sample_h else h_prob
    v_prob = visible_probs(st, h)
    v_next = sample_visible(st, v_prob, rng) if sample_v else v_prob
    return v_next, v_prob, h, h_prob


# ---------------------------------------------------------------------------
# CD-k update  (rbm.py:194-227, Appendix A.1)
# ---------------------------------------------------------------------------
def _lr_mom(st: RBMState, epoch: int):
    lr = st.lr / (1 + 0.01 * epoch) if st.dynamic_lr else st.lr     # rbm.py:194
    mom = st.momentum if epoch <= 5 else st.final_momentum           # rbm.py:195
    return lr, mom


def cd_statistics(st: RBMState, data: np.ndarray, CD: int, rng):
    """Positive/negative phase of rbm.py:199-209 without the parameter update.

    Returns un-normalised sufficient statistics -- this is also what one data-parallel
    rank contributes before the all-reduce.
    """
    data = np.asarray(data, F32)
    pos_h = forward(st, data)                                  # :199
    pos_assoc = (data.T @ pos_h).astype(F32)                   # :200
    h = _bern(pos_h, rng.uniform(pos_h.shape))         # :203
    v = v_prob = h_prob = None
    for _ in range(int(CD)):                                   # :204
        v_prob = visible_probs(st, h)                          # :205
        v = sample_visible(st, v_prob, rng)                    # :206
        h_prob = forward(st, v)                                # :207
        h = _bern(h_prob, rng.uniform(h_prob.shape))   # :208 (last draw discarded)
    neg_assoc = (v.T @ h_prob).astype(F32)                     # :209
    return dict(
        pos_assoc=pos_assoc, neg_assoc=neg_assoc,
        pos_h_sum=pos_h.sum(0, dtype=F32), neg_h_sum=h_prob.sum(0, dtype=F32),
        data_sum=data.sum(0, dtype=F32), v_sum=v.sum(0, dtype=F32),
        sq_err=((data - v_prob) ** 2).astype(F32), n=data.shape[0],
        pos_h=pos_h, h_prob=h_prob, v=v, v_prob=v_prob,
    )


def apply_cd_update(st: RBMState, s: dict, lr: float, mom: float, bsz: int, sparsity: bool):
    """rbm.py:212-224 given (possibly all-reduced) statistics."""
    lr32, mom32, n32 = F32(lr), F32(mom), F32(bsz)
    st.W_m *= mom32
    st.W_m += lr32 * ((s["pos_assoc"] - s["neg_assoc"]) / n32 - F32(st.weight_decay) * st.W)   # :212
    st.W += st.W_m                                                                             # :213
    st.hb_m *= mom32
    st.hb_m += lr32 * (s["pos_h_sum"] - s["neg_h_sum"]) / n32                                  # :216
    if sparsity:
        Q = (s["pos_h_sum"] / n32).astype(F32)                                                 # :218 mean(0)
        st.hb_m += F32(-lr) * (Q - F32(st.sparsity_factor))                                    # :219
    st.hid_bias += st.hb_m                                                                     # :220
    st.vb_m *= mom32
    st.vb_m += lr32 * (s["data_sum"] - s["v_sum"]) / n32                                       # :223
    st.vis_bias += st.vb_m                                                                     # :224


def train_epoch(st: RBMState, data: np.ndarray, epoch: int, CD: int, rng) -> np.float32:
    """rbm.py:181-227.  One CD-k update on one mini-batch; returns the MSE loss."""
    lr, mom = _lr_mom(st, epoch)
    s = cd_statistics(st, data, CD, rng)
    apply_cd_update(st, s, lr, mom, s["n"], st.sparsity)
    return F32(s["sq_err"].mean(dtype=F32))                                                    # :226


def train_epoch_sharded(st: RBMState, shards: Sequence[np.ndarray], epoch: int, CD: int,
                        rngs: Sequence) -> np.float32:
    """Data-parallel semantics: every rank computes statistics on its
    rows with the *same* pre-update parameters, statistics are summed (the all-reduce),
    and every replica applies the identical update with 1/global_batch."""
    lr, mom = _lr_mom(st, epoch)
    parts = [cd_statistics(st, d, CD, r) for d, r in zip(shards, rngs)]
    tot = {k: sum((p[k] for p in parts[1:]), parts[0][k].copy())
           for k in ("pos_assoc", "neg_assoc", "pos_h_sum", "neg_h_sum", "data_sum", "v_sum")}
    n = sum(p["n"] for p in parts)
    sq = sum(F32(p["sq_err"].sum(dtype=F32)) for p in parts)
    apply_cd_update(st, tot, lr, mom, n, st.sparsity)
    return F32(sq / F32(n * st.num_visible))


# ---------------------------------------------------------------------------
# schedules  (rbm.py:229-238)
# ---------------------------------------------------------------------------
def lin_schedule(t, t_max, start, end) -> float:
    if t_max <= 1:
        return float(end)
    alpha = min(max(t / (t_max - 1), 0.0), 1.0)
    return float(start + (end - start) * alpha)


def hot_steps(n_steps, hot_frac) -> int:
    return int(max(0, min(n_steps, round(hot_frac * n_steps))))


def nmf_schedule(n_steps, T0=3.0, T1=1.0, sigma0=0.9, sharpen_last=3, T_cold_plus=0.9, eta0=None):
    """Per-step (T_t, sigma_t, eta_t) of rbm.py:337-341,362 as host scalars."""
    out = []
    n = int(n_steps)
    for t in range(n):
        Tt = lin_schedule(t, n, T0, T1)
        if (n - t) <= max(1, int(sharpen_last)):
            Tt = T_cold_plus
        frac = max(0.0, 1.0 - (t / max(1, n - 1)))
        out.append((Tt, sigma0 * frac, (eta0 * frac) if eta0 is not None else 0.0))
    return out


# ---------------------------------------------------------------------------
# chains  (rbm.py:240-400, Appendix A.2/A.3)
# ---------------------------------------------------------------------------
def noisy_meanfield_annealed(st: RBMState, v_known, known_mask, rng, n_steps=72, T0=3.0, T1=1.0,
                             sigma0=0.9, hot_frac=0.7, sharpen_last=3, T_cold_plus=0.9):
    """rbm.py:332-367."""
    v_known = np.asarray(v_known, F32)
    km = np.asarray(known_mask, F32)
    v = v_known * km + (F32(1) - km) * rng.uniform(v_known.shape)              # :333
    n = int(n_steps)
    for t in range(n):
        Tt = lin_schedule(t, n, T0, T1)                                       # :338
        if (n - t) <= max(1, int(sharpen_last)):                              # :339
            Tt = T_cold_plus
        sig_t = sigma0 * max(0.0, 1.0 - (t / max(1, n - 1)))                   # :341
        h_logits = ((v @ st.W) + st.hid_bias) / _temp(Tt)                      # :344
        if sig_t > 0:
            h_logits = h_logits + rng.normal(h_logits.shape) * F32(sig_t)      # :346
        h_prob = sigmoid(h_logits)                                            # :347
        v_logits = ((h_prob @ st.W.T) + st.vis_bias) / _temp(Tt)               # :350
        if sig_t > 0:
            v_logits = v_logits + rng.normal(v_logits.shape) * F32(sig_t)      # :352
        v_logits = v_logits.astype(F32)
        v_prob = _apply_groups(st, sigmoid(v_logits), v_logits)               # :354-356
        if st.mu_pull is not None:                                            # :359-363
            mu = np.asarray(st.mu_pull["mu_k"], F32)
            Dz = mu.shape[1]
            eta0 = float(st.mu_pull.get("eta0", 0.15))
            eta_t = eta0 * max(0.0, 1.0 - (t / max(1, n - 1)))
            v_prob[:, :Dz] = F32(1 - eta_t) * v_prob[:, :Dz] + F32(eta_t) * mu
        v = (v_prob * (F32(1) - km) + v_known * km).astype(F32)               # :365
    return v


def conditional_gibbs(st: RBMState, v_known, known_mask, rng, n_steps=30, sample_h=False, sample_v=False):
    """rbm.py:391-400 (note the final un-clamped pass)."""
    v_known = np.asarray(v_known, F32)
    km = np.asarray(known_mask, F32)
    v = v_known * km + (F32(1) - km) * rng.uniform(v_known.shape)              # :392
    for _ in range(int(n_steps)):
        h_prob = forward(st, v)                                               # :394
        h = _bern(h_prob, rng.uniform(h_prob.shape)) if sample_h else h_prob
        v_prob = visible_probs(st, h)                                         # :396
        v = (v_prob * (F32(1) - km) + v_known * km).astype(F32)               # :397
        if sample_v:
            v = (sample_visible(st, v, rng) * (F32(1) - km) + v_known * km).astype(F32)   # :399
    return visible_probs(st, forward(st, v))                                  # :400


def conditional_gibbs_annealed(st: RBMState, v_known, known_mask, rng, n_steps=40, T0=2.5, T1=1.0,
                               sample_h_until=20, sample_v_every=0, final_meanfield=True):
    """rbm.py:270-298."""
    v_known = np.asarray(v_known, F32)
    km = np.asarray(known_mask, F32)
    v = v_known * km + (F32(1) - km) * rng.uniform(v_known.shape)              # :271
    hot = int(max(0, min(n_steps, sample_h_until)))                           # :273
    n = int(n_steps)
    for t in range(n):
        Tt = lin_schedule(t, n, T0, T1)
        if (n - t) <= 3:
            Tt = min(0.9, Tt)                                                 # :278-279
        h_prob = forward(st, v, T=Tt)
        h = _bern(h_prob, rng.uniform(h_prob.shape)) if t < hot else h_prob   # :282
        v_prob = visible_probs(st, h, T=Tt)
        if (t < hot) and (sample_v_every > 0) and (t % sample_v_every == 0):
            v_new = sample_visible(st, v_prob, rng)                           # :286
        else:
            v_new = v_prob
        v = (v_new * (F32(1) - km) + v_known * km).astype(F32)                # :291
    if final_meanfield:
        h_prob = forward(st, v, T=1.0)
        v = (visible_probs(st, h_prob, T=1.0) * (F32(1) - km) + v_known * km).astype(F32)   # :295-296
    return v


# ---------------------------------------------------------------------------
# clamped CD  (rbm.py:438-483, Appendix A.4)
# ---------------------------------------------------------------------------
def clamped_statistics(st: RBMState, v_known, known_mask, rng, CD=1, cond_init_steps=50, sample_h=True,
                       sample_v=False, reclamp_negative=True, use_noisy_init=True):
    v_known = np.asarray(v_known, F32)
    km = np.asarray(known_mask, F32)
    if use_noisy_init:                                                        # :443-448
        v_plus = noisy_meanfield_annealed(st, v_known, km, rng, n_steps=max(10, int(cond_init_steps)),
                                          T0=3.0, T1=1.0, sigma0=0.9, hot_frac=0.7, sharpen_last=2,
                                          T_cold_plus=0.9)
    else:                                                                     # :450-453
        v_plus = conditional_gibbs(st, v_known, km, rng, n_steps=cond_init_steps,
                                   sample_h=sample_h, sample_v=sample_v)
    h_plus = forward(st, v_plus)                                              # :455
    pos_assoc = (v_plus.T @ h_plus).astype(F32)                               # :456
    v_neg = v_plus.copy()
    for _ in range(int(CD)):                                                  # :460-469
        h_prob = forward(st, v_neg)
        h = _bern(h_prob, rng.uniform(h_prob.shape)) if sample_h else h_prob
        v_prob = visible_probs(st, h)
        if reclamp_negative:
            v_neg = (v_prob * (F32(1) - km) + v_known * km).astype(F32)
        else:
            v_neg = v_prob
        if sample_v:
            v_neg = sample_visible(st, v_neg, rng)
    h_neg = forward(st, v_neg)                                                # :471
    neg_assoc = (v_neg.T @ h_neg).astype(F32)                                 # :472
    return dict(pos_assoc=pos_assoc, neg_assoc=neg_assoc,
                pos_h_sum=h_plus.sum(0, dtype=F32), neg_h_sum=h_neg.sum(0, dtype=F32),
                data_sum=v_plus.sum(0, dtype=F32), v_sum=v_neg.sum(0, dtype=F32),
                sq_err=((v_plus - v_neg) ** 2).astype(F32), n=v_known.shape[0],
                v_plus=v_plus, v_neg=v_neg, h_plus=h_plus, h_neg=h_neg)


def train_epoch_clamped(st: RBMState, v_known, known_mask, epoch: int, rng, CD=1, cond_init_steps=50,
                        sample_h=True, sample_v=False, reclamp_negative=True, aux_lr_mult=0.3,
                        use_noisy_init=True) -> np.float32:
    """rbm.py:403-483."""
    lr, mom = _lr_mom(st, epoch)
    s = clamped_statistics(st, v_known, known_mask, rng, CD, cond_init_steps, sample_h, sample_v,
                           reclamp_negative, use_noisy_init)
    # :475-481 -- same algebra as the free update with lr -> aux_lr_mult*lr and no sparsity
    apply_cd_update(st, s, aux_lr_mult * lr, mom, s["n"], sparsity=False)
    return F32(s["sq_err"].mean(dtype=F32))                                   # :483


# ---------------------------------------------------------------------------
# iDBN  (idbn.py:195-204, 307-359)
# ---------------------------------------------------------------------------
def idbn_train_batch(layers: List[RBMState], v: np.ndarray, epoch: int, cd_k: int, rng) -> List[np.float32]:
    """idbn.py:200-204: interleaved greedy update -- each layer is updated on every batch
    and feeds the next layer with probabilities computed from its UPDATED weights."""
    losses = []
    v = np.asarray(v, F32).reshape(v.shape[0], -1)
    for st in layers:
        losses.append(train_epoch(st, v, epoch, cd_k, rng))                   # :202
        v = forward(st, v)                                                    # :203
    return losses


def idbn_represent(layers: List[RBMState], x, upto_layer=None):
    """idbn.py:319-323."""
    v = np.asarray(x, F32).reshape(x.shape[0], -1)
    L = len(layers) if upto_layer is None else max(0, min(len(layers), int(upto_layer)))
    for i in range(L):
        v = forward(layers[i], v)
    return v


def idbn_decode(layers: List[RBMState], top):
    """idbn.py:356-359."""
    cur = np.asarray(top, F32)
    for st in reversed(layers):
        cur = backward(st, cur)
    return cur


def idbn_reconstruct(layers: List[RBMState], x):
    """idbn.py:336-344."""
    return idbn_decode(layers, idbn_represent(layers, x))


# ---------------------------------------------------------------------------
# iMDBN  (imdbn.py:216-292, 386-488, 553-639)
# ---------------------------------------------------------------------------
def init_joint_bias_from_data(img_layers, joint: RBMState, batches, num_labels: int, n_batches: int = 10):
    """imdbn.py:237-292.  ``batches`` = list of (img[B,D], y_onehot[B,K]).  Returns z_class_mean."""
    Dz = joint.num_visible - num_labels
    K = num_labels
    sum_z = None
    n = 0
    class_counts = np.zeros(K, F32)
    for b, (imgs, lbls) in enumerate(batches):
        if b >= n_batches:
            break
        z = idbn_represent(img_layers, imgs)
        sum_z = z.sum(0, dtype=F32) if sum_z is None else (sum_z + z.sum(0, dtype=F32))
        n += z.shape[0]
        class_counts += np.asarray(lbls, F32).sum(0, dtype=F32)
    if n == 0:
        return None
    mean_z = np.clip(sum_z / F32(n), F32(1e-4), F32(1 - 1e-4)).astype(F32)    # :256
    priors = class_counts / F32(max(1, class_counts.sum()))                   # :257
    priors = ((priors + F32(1e-6)) / (priors.sum(dtype=F32) + F32(1e-6 * K))).astype(F32)   # :258
    zcm = np.zeros((K, Dz), F32)
    cnt = np.zeros(K, F32)
    for b, (imgs, lbls) in enumerate(batches):
        if b >= n_batches:
            break
        z = idbn_represent(img_layers, imgs)
        y_idx = np.asarray(lbls).argmax(1)
        for k in range(K):
            m = (y_idx == k)
            if m.any():
                zcm[k] += z[m].sum(0, dtype=F32)                              # :276
                cnt[k] += F32(m.sum())
    for k in range(K):
        if cnt[k] > 0:
            zcm[k] /= cnt[k]                                                  # :282
        else:
            zcm[k] = mean_z
    joint.vis_bias[:Dz] = np.log(mean_z) - np.log1p(-mean_z)                  # :291
    joint.vis_bias[Dz:Dz + K] = np.log(priors)                                # :292
    return zcm


def cross_reconstruct(img_layers, joint: RBMState, z_img, y_onehot, steps: int, rng,
                      z_class_mean=None, Kbuf: int = 5):
    """imdbn.py:410-488.  The reference's best-of-K is inert (RBM has no ``free_energy``,
    imdbn.py:455-470): candidate energies are all zero, argmin picks index 0, so the result
    is always the main chain; the ``Kbuf-1`` refinement passes still consume draws."""
    z_img = np.asarray(z_img, F32)
    y = np.asarray(y_onehot, F32)
    B, Dz = z_img.shape
    K = y.shape[1]
    V = Dz + K
    v_known = np.zeros((B, V), F32)
    km = np.zeros((B, V), F32)
    v_known[:, :Dz] = z_img
    km[:, :Dz] = 1.0
    v_i2t = conditional_gibbs(joint, v_known, km, rng, n_steps=steps, sample_h=False, sample_v=False)  # :424
    p_y = v_i2t[:, Dz:].copy()
    v_known[:] = 0
    km[:] = 0
    v_known[:, Dz:] = y
    km[:, Dz:] = 1.0
    if z_class_mean is not None:                                              # :436-442
        joint.mu_pull = {"mu_k": np.asarray(z_class_mean, F32)[y.argmax(1)], "eta0": 0.15}
    else:
        joint.mu_pull = None
    v_chain = noisy_meanfield_annealed(joint, v_known, km, rng, n_steps=steps, T0=3.0, T1=1.0, sigma0=0.9,
                                       hot_frac=0.7, sharpen_last=3, T_cold_plus=0.9)            # :445
    last = v_chain
    for _ in range(Kbuf - 1):                                                 # :460-470 (discarded)
        last = noisy_meanfield_annealed(joint, last, km, rng, n_steps=1, T0=0.9, T1=0.9, sigma0=0.0,
                                        hot_frac=0.0, sharpen_last=0, T_cold_plus=0.9)
    joint.mu_pull = None                                                      # :476
    z_from_y = v_chain[:, :Dz]
    img_from_txt = idbn_decode(img_layers, z_from_y)                          # :487
    return img_from_txt, p_y


def joint_represent(img_layers, joint: RBMState, img, y):
    """imdbn.py:501-506."""
    z = idbn_represent(img_layers, img)
    return forward(joint, np.concatenate([z, np.asarray(y, F32)], axis=1))


def train_joint_batch(img_layers, joint: RBMState, img, y, epoch: int, b_idx: int, rng, joint_cd: int,
                      aux_cond_steps: int, cross_steps: int, z_class_mean=None, warmup_epochs: int = 8):
    """One iteration of the batch loop imdbn.py:553-639; returns the online metric terms."""
    img = np.asarray(img, F32).reshape(img.shape[0], -1)
    y = np.asarray(y, F32)
    z_img = idbn_represent(img_layers, img)                                   # :558
    v_plus = np.concatenate([z_img, y], axis=1)                               # :559
    B, Dz = z_img.shape
    K = y.shape[1]
    V = Dz + K
    loss_cd = None

    def yclamp():
        vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
        vk[:, Dz:] = y; km[:, Dz:] = 1.0
        return vk, km

    if epoch < warmup_epochs:                                                 # :566-579
        for _ in range(2):
            vk, km = yclamp()
            train_epoch_clamped(joint, vk, km, epoch, rng, CD=1, cond_init_steps=aux_cond_steps,
                                sample_h=False, sample_v=False, aux_lr_mult=0.3, use_noisy_init=True)
    else:                                                                     # :582-612
        loss_cd = train_epoch(joint, v_plus, epoch, joint_cd, rng)
        vk, km = yclamp()
        train_epoch_clamped(joint, vk, km, epoch, rng, CD=1, cond_init_steps=aux_cond_steps,
                            sample_h=False, sample_v=False, reclamp_negative=False, aux_lr_mult=0.3,
                            use_noisy_init=True)
        if (b_idx % 50) == 0:
            vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
            vk[:, :Dz] = z_img; km[:, :Dz] = 1.0
            train_epoch_clamped(joint, vk, km, epoch, rng, CD=1, cond_init_steps=aux_cond_steps,
                                sample_h=False, sample_v=False, reclamp_negative=False, aux_lr_mult=0.3,
                                use_noisy_init=True)
    img_from_txt, p_y = cross_reconstruct(img_layers, joint, z_img, y, cross_steps, rng, z_class_mean)  # :616
    gt = y.argmax(1)
    pred = p_y.argmax(1)
    top3 = np.argsort(-p_y, axis=1, kind="stable")[:, :min(3, K)]
    pc = np.clip(p_y, F32(1e-6), F32(1 - 1e-6)).astype(F32)
    onehot = np.eye(K, dtype=F32)[gt]
    ce = F32(-(onehot * np.log(pc) + (F32(1) - onehot) * np.log(F32(1) - pc)).sum(dtype=F32))          # :624
    mse = F32(((img_from_txt - img) ** 2).sum(dtype=F32))                                              # :630
    return dict(loss_cd=loss_cd, n=B, top1=int((pred == gt).sum()),
                top3=int((top3 == gt[:, None]).any(1).sum()), ce_sum=float(ce), mse_sum=float(mse),
                img_from_txt=img_from_txt, p_y=p_y)


# ---------------------------------------------------------------------------
# free energy (imdbn/utils/energy_utils.py:19-28) -- spec for the opt-in live best-of-K
# ---------------------------------------------------------------------------
# ---- iMDBN_BiModal (imdbn_bimodal.py) ------------------------------------------------------------
def bimodal_init_joint_bias(mod1_layers, mod2_layers, joint0: RBMState, batches, n_batches: int = 10):
    """imdbn_bimodal.py:617-645: logit of the clamped mean latent of each modality -> first joint layer's vis_bias."""
    s1 = s2 = None
    n = 0
    for b, (m1, m2) in enumerate(batches):
        if b >= n_batches:
            break
        z1 = idbn_represent(mod1_layers, np.asarray(m1, F32).reshape(len(m1), -1))
        z2 = idbn_represent(mod2_layers, np.asarray(m2, F32).reshape(len(m2), -1))
        s1 = z1.sum(0, dtype=F32) if s1 is None else (s1 + z1.sum(0, dtype=F32)).astype(F32)
        s2 = z2.sum(0, dtype=F32) if s2 is None else (s2 + z2.sum(0, dtype=F32)).astype(F32)
        n += z1.shape[0]
    if n == 0:
        return
    Dz1 = s1.shape[0]
    m1 = np.clip((s1 / F32(n)).astype(F32), F32(1e-4), F32(1 - 1e-4)).astype(F32)
    m2 = np.clip((s2 / F32(n)).astype(F32), F32(1e-4), F32(1 - 1e-4)).astype(F32)
    joint0.vis_bias[:Dz1] = (np.log(m1) - np.log1p(-m1)).astype(F32)
    joint0.vis_bias[Dz1:] = (np.log(m2) - np.log1p(-m2)).astype(F32)


def bimodal_cross_reconstruct(mod1_layers, mod2_layers, joint0: RBMState, z1, z2, steps: int, rng):
    """imdbn_bimodal.py:648-693: Gibbs completion (sampled hidden units) in both directions, then decode."""
    z1 = np.asarray(z1, F32); z2 = np.asarray(z2, F32)
    B, Dz1 = z1.shape
    V = Dz1 + z2.shape[1]
    vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
    vk[:, :Dz1] = z1; km[:, :Dz1] = 1.0
    v12 = conditional_gibbs(joint0, vk, km, rng, n_steps=steps, sample_h=True, sample_v=False)
    vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
    vk[:, Dz1:] = z2; km[:, Dz1:] = 1.0
    v21 = conditional_gibbs(joint0, vk, km, rng, n_steps=steps, sample_h=True, sample_v=False)
    return idbn_decode(mod1_layers, v21[:, :Dz1]), idbn_decode(mod2_layers, v12[:, Dz1:])


def bimodal_represent(mod1_layers, mod2_layers, joint_layers, m1, m2):
    """imdbn_bimodal.py:696-709."""
    h = np.concatenate([idbn_represent(mod1_layers, np.asarray(m1, F32).reshape(len(m1), -1)),
                        idbn_represent(mod2_layers, np.asarray(m2, F32).reshape(len(m2), -1))], axis=1)
    for st in joint_layers:
        h = forward(st, h)
    return h


def bimodal_train_joint_batch(mod1_layers, mod2_layers, joint_layers, m1, m2, epoch: int, rng, joint_cd: int,
                              aux_cond_steps: int, cross_steps: int, warmup_epochs: int = 8):
    """One iteration of the batch loop imdbn_bimodal.py:741-829; returns the online metric terms."""
    v1 = np.asarray(m1, F32).reshape(len(m1), -1)
    v2 = np.asarray(m2, F32).reshape(len(m2), -1)
    z1 = idbn_represent(mod1_layers, v1)
    z2 = idbn_represent(mod2_layers, v2)
    B, Dz1 = z1.shape
    V = Dz1 + z2.shape[1]
    first = joint_layers[0]
    loss_cd = None

    def clamp(z, lo):
        vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
        vk[:, lo:lo + z.shape[1]] = z; km[:, lo:lo + z.shape[1]] = 1.0
        return vk, km

    if epoch < warmup_epochs:                                                 # :750-781
        for _ in range(2):
            for z, lo in ((z1, 0), (z2, Dz1)):
                vk, km = clamp(z, lo)
                train_epoch_clamped(first, vk, km, epoch, rng, CD=3, cond_init_steps=aux_cond_steps,
                                    sample_h=True, sample_v=False, aux_lr_mult=0.3, use_noisy_init=True)
    else:                                                                     # :783-820
        cur = np.concatenate([z1, z2], axis=1)
        for li, st in enumerate(joint_layers):
            loss = train_epoch(st, cur, epoch, joint_cd, rng)
            if li == 0:
                loss_cd = loss
            cur = forward(st, cur)
        for z, lo in ((z1, 0), (z2, Dz1)):
            vk, km = clamp(z, lo)
            train_epoch_clamped(first, vk, km, epoch, rng, CD=3, cond_init_steps=aux_cond_steps,
                                sample_h=True, sample_v=False, reclamp_negative=False, aux_lr_mult=0.3,
                                use_noisy_init=True)
    r1, r2 = bimodal_cross_reconstruct(mod1_layers, mod2_layers, first, z1, z2, cross_steps, rng)
    return dict(loss_cd=loss_cd, n=B, mse1_sum=float(F32(((r1 - v1) ** 2).sum(dtype=F32))),
                mse2_sum=float(F32(((r2 - v2) ** 2).sum(dtype=F32))), mod1_from_mod2=r1, mod2_from_mod1=r2)


def free_energy(st: RBMState, v: np.ndarray) -> np.ndarray:
    v = np.asarray(v, F32)
    wx_b = (v @ st.W) + st.hid_bias
    softplus = np.logaddexp(F32(0), wx_b).astype(F32)
    return (-(v @ st.vis_bias) - softplus.sum(1, dtype=F32)).astype(F32)
