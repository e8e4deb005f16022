"""Parity cases driven through the PRODUCT's class API (imdbn.models.*), device-agnostic.

The same bodies run (a) on CPU with the oracle-backed test double installed -> validates host
logic, schedules and draw order against the reference fixtures, and (b) on the MI355X with the
real HipEngine -> the parity tests proper.  They read like the reference's own usage:
``RBM(...).train_epoch(...)``, ``iDBN(...).train()``, ``iMDBN(...).train_joint()``.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset

from golden_utils import Fixture, assert_close, init_W
from imdbn import engine as E
from imdbn.models import RBM, iDBN, iMDBN, iMDBN_BiModal

F32 = np.float32
KEYS = ("W", "hid_bias", "vis_bias", "W_m", "hb_m", "vb_m")


def T(a, dev):
    return torch.from_numpy(np.array(a, dtype=F32, copy=True)).to(dev)     # copy: never alias fixture arrays


def N(t):
    return t.detach().float().cpu().numpy()


def set_params(r, dev, W, hb=None, vb=None):
    r.to(dev)
    r.W.data = T(W, dev)
    r.hid_bias.data = T(hb, dev) if hb is not None else torch.zeros(W.shape[1], device=dev)
    r.vis_bias.data = T(vb, dev) if vb is not None else torch.zeros(W.shape[0], device=dev)
    r.W_m = torch.zeros_like(r.W.data)
    r.hb_m = torch.zeros_like(r.hid_bias.data)
    r.vb_m = torch.zeros_like(r.vis_bias.data)
    return r


def check_state(r, fx, prefix, rel):
    for k in KEYS:
        if prefix + k in fx.a:
            assert_close(N(getattr(r, k)), fx[prefix + k], rel=rel, what=prefix + k, atol=3e-6)


def loader(X, Y, B):
    return DataLoader(TensorDataset(torch.from_numpy(X), torch.from_numpy(Y)), batch_size=B, shuffle=False)


# -------------------------------------------------------------------------------------------------
def case_c1(dev, rel=1e-4):
    """BASELINE configs[0]: RBM 784<->256, CD-1, batch 32, 20 updates (north_star tolerance 1e-4)."""
    fx = Fixture("c1_rbm784x256_cd1.npz")
    m = fx.meta
    s = fx.stream()
    r = RBM(m["V"], m["H"], m["lr"], m["wd"], m["mom"], dynamic_lr=True, final_momentum=m["final_momentum"])
    set_params(r, dev, init_W(s, m["V"], m["H"]))
    X = (s.uniform((640, m["V"])) > 0.5).astype(F32)
    losses = []
    with E.use_rng(E.ReplayRng(s)):
        for i in range(m["updates"]):
            losses.append(r.train_epoch(T(X[32 * i:32 * i + 32], dev), m["epoch"], 10, CD=m["CD"]))
            if i == 0:
                assert_close(N(r.W)[::16], fx["a1_W_rows"], rel=1e-5, what="W after 1 update")
                assert_close(N(r.W_m)[::16], fx["a1_W_m_rows"], rel=5e-5, what="W_m after 1 update", atol=1e-7)
                assert_close(N(r.hid_bias), fx["a1_hid_bias"], rel=5e-5, what="hid_bias after 1 update", atol=1e-7)
                assert_close(N(r.vis_bias), fx["a1_vis_bias"], rel=5e-5, what="vis_bias after 1 update", atol=1e-7)
    assert_close(np.array([float(l) for l in losses], F32), fx["losses"], rel=1e-5, what="losses")
    assert_close(N(r.W), fx["W"], rel=rel, what="W after 20 updates")
    for k in ("hid_bias", "vis_bias", "hb_m", "vb_m"):
        assert_close(N(getattr(r, k)), fx[k], rel=rel, what=k, atol=1e-6)
    assert_close(N(r.W_m)[::16], fx["W_m_rows"], rel=rel, what="W_m rows", atol=1e-7)


# -------------------------------------------------------------------------------------------------
def case_joint_small(dev, rel=1e-4):
    """Every RBM method on a 96+8 <-> 40 RBM with one softmax group (same order as the generator)."""
    fx = Fixture("joint_small_rbm104x40.npz")
    m = fx.meta
    V, H, B, Dz = m["V"], m["H"], m["B"], m["Dz"]
    s = fx.stream()
    W0 = init_W(s, V, H)
    hb0 = (s.normal((H,)) * F32(0.1)).astype(F32)
    vb0 = (s.normal((V,)) * F32(0.1)).astype(F32)
    z = s.uniform((B, Dz)).astype(F32)
    y = np.eye(8, dtype=F32)[fx["yi"]]
    data = np.concatenate([(z > 0.5).astype(F32), y], 1)
    data_real = np.concatenate([z, y], 1)
    mu = s.uniform((B, Dz)).astype(F32)

    def fresh(**kw):
        r = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, softmax_groups=[(96, 104)], **kw)
        return set_params(r, dev, W0, hb0, vb0)

    def clamp(which):
        vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
        if which == "y":
            vk[:, Dz:] = y; km[:, Dz:] = 1
        else:
            vk[:, :Dz] = z; km[:, :Dz] = 1
        return T(vk, dev), T(km, dev)

    with E.use_rng(E.ReplayRng(s)):
        r = fresh()
        td = T(data_real, dev)
        assert_close(N(r.forward(td)), fx["fwd_T1"], 1e-5, "forward T=1")
        assert_close(N(r.forward(td, T=2.5)), fx["fwd_T25"], 1e-5, "forward T=2.5")
        th = r.forward(td)
        assert_close(N(r.visible_probs(th)), fx["vis_T1"], 1e-5, "visible_probs")
        assert_close(N(r.visible_probs(th, T=0.7)), fx["vis_T07"], 1e-5, "visible_probs T=.7")
        assert_close(N(r.backward(th, return_logits=True)), fx["bwd_logits"], 1e-5, "backward logits", atol=1e-6)
        assert_close(N(r.backward(th)), fx["bwd"], 1e-5, "backward")
        np.testing.assert_array_equal(N(r.sample_visible(r.visible_probs(th))), fx["sample_visible"])
        np.testing.assert_array_equal(N(r.backward_sample(th)), fx["backward_sample"])
        g = r.gibbs_step(T(data, dev))
        np.testing.assert_array_equal(N(g[0]), fx["gibbs_v_next"])
        assert_close(N(g[1]), fx["gibbs_v_prob"], 1e-5, "gibbs v_prob")
        np.testing.assert_array_equal(N(g[2]), fx["gibbs_h"])
        assert_close(N(g[3]), fx["gibbs_h_prob"], 1e-5, "gibbs h_prob")

        r = fresh(sparsity=True, sparsity_factor=0.1)
        loss = r.train_epoch(T(data, dev), 7, 20, CD=2)
        assert_close(float(loss), fx["te_loss"], 1e-5, "train_epoch CD2 loss")
        check_state(r, fx, "te_", rel)

        r = fresh()
        l1 = r.train_epoch(T(data_real, dev), 0, 20, CD=1)
        l2 = r.train_epoch(T(data_real, dev), 1, 20, CD=1)
        assert_close(np.array([float(l1), float(l2)], F32), fx["ter_loss"], 1e-5, "train_epoch real loss")
        check_state(r, fx, "ter_", rel)

        r = fresh()
        vk, km = clamp("z")
        assert_close(N(r.conditional_gibbs(vk, km, n_steps=10)), fx["cg_plain"], 5e-5, "conditional_gibbs")
        assert_close(N(r.conditional_gibbs(vk, km, n_steps=5, sample_h=True, sample_v=True)), fx["cg_sampled"], 5e-5,
                     "conditional_gibbs sampled")
        assert_close(N(r.conditional_gibbs_annealed(vk, km, n_steps=12, T0=2.5, T1=1.0, sample_h_until=6,
                                                    sample_v_every=2, final_meanfield=True)), fx["cga"], 5e-5, "cga")
        assert_close(N(r.conditional_gibbs_annealed(vk, km, n_steps=6, sample_h_until=0, final_meanfield=False)),
                     fx["cga_nofinal"], 5e-5, "cga no final")
        vk, km = clamp("y")
        assert_close(N(r.noisy_meanfield_annealed(vk, km, n_steps=20)), fx["nmf20"], 5e-5, "nmf20")
        r._mu_pull = {"mu_k": T(mu, dev), "eta0": 0.15}
        assert_close(N(r.noisy_meanfield_annealed(vk, km, n_steps=20, sharpen_last=3)), fx["nmf20_mu"], 5e-5, "nmf20 mu")
        assert_close(N(r.noisy_meanfield_annealed(T(fx["nmf20_mu"], dev), km, n_steps=1, T0=0.9, T1=0.9, sigma0=0.0,
                                                  hot_frac=0.0, sharpen_last=0, T_cold_plus=0.9)), fx["nmf1_mu"], 5e-5,
                     "nmf 1-step refinement")
        r._mu_pull = None

        for tag, kw in (
            ("tc_noisy_reclamp", dict(CD=1, cond_init_steps=12, sample_h=False, sample_v=False, reclamp_negative=True,
                                      aux_lr_mult=0.3, use_noisy_init=True)),
            ("tc_noisy_noreclamp", dict(CD=1, cond_init_steps=4, sample_h=False, sample_v=False, reclamp_negative=False,
                                        aux_lr_mult=0.3, use_noisy_init=True)),
            ("tc_gibbs_sampled", dict(CD=3, cond_init_steps=6, sample_h=True, sample_v=True, reclamp_negative=True,
                                      aux_lr_mult=0.5, use_noisy_init=False)),
            ("tc_defaults", dict()),
        ):
            r = fresh()
            vk, km = clamp("y")
            loss = r.train_epoch_clamped(vk, km, 9, 20, **kw)
            assert_close(float(loss), fx[tag + "_loss"], 1e-4, tag + " loss")
            check_state(r, fx, tag + "_", rel)
    assert s.exhausted_cat()
    got_log = ";".join(f"{k}{'x'.join(map(str, sh))}" for k, sh in s.log)
    assert got_log == m["draw_log"], "draw order differs from the reference's (Appendix B)"


# -------------------------------------------------------------------------------------------------
def case_idbn_small(dev, rel=1e-4):
    fx = Fixture("idbn_small_100_40_20.npz")
    m = fx.meta
    s = fx.stream()
    sizes, Nn, B = m["sizes"], m["N"], m["B"]
    X = (s.uniform((Nn, sizes[0])) > 0.8).astype(F32)
    dl = loader(X, np.zeros((Nn, 1), F32), B)
    d = iDBN(sizes, dict(m["params"]), dl, dl, torch.device(dev))
    for i, r in enumerate(d.layers):
        set_params(r, dev, init_W(s, sizes[i], sizes[i + 1]))
    with E.use_rng(E.ReplayRng(s)):
        d.train(m["epochs"])
    losses = torch.cat(d.loss_history).numpy()
    assert_close(losses, fx["losses"], 5e-5, "losses")
    for i, r in enumerate(d.layers):
        check_state(r, fx, f"L{i}_", rel)
    xt = torch.from_numpy(X[:8])
    assert_close(N(d.represent(xt)), fx["represent"], 5e-5, "represent")
    assert_close(N(d.represent(xt, upto_layer=1)), fx["represent_l1"], 5e-5, "represent l1")
    assert_close(N(d.reconstruct(xt)), fx["reconstruct"], 5e-5, "reconstruct")
    assert_close(N(d.decode(d.represent(xt))), fx["decode"], 5e-5, "decode")
    assert len(s.log) == m["draw_log_len"]
    return d


# -------------------------------------------------------------------------------------------------
def case_imdbn_small(dev, rel=3e-4):
    fx = Fixture("imdbn_small_100_40_20_j16.npz")
    m = fx.meta
    s = fx.stream()
    sizes, JH, K, B, NB = m["sizes"], m["joint_hidden"], m["K"], m["B"], m["NB"]
    Nn = B * NB
    yi = fx["yi"]
    proto = (s.uniform((K, 100)) > 0.7).astype(F32)
    flip = (s.uniform((Nn, 100)) > 0.9).astype(F32)
    X = np.abs(proto[yi] - flip).astype(F32)
    Y = np.eye(K, dtype=F32)[yi]
    dl = loader(X, Y, B)
    mdl = iMDBN(sizes, JH, params=dict(m["params"]), dataloader=dl, val_loader=dl, device=torch.device(dev), num_labels=K)
    for i, r in enumerate(mdl.image_idbn.layers):
        set_params(r, dev, init_W(s, sizes[i], sizes[i + 1]))
    set_params(mdl.joint_rbm, dev, init_W(s, sizes[-1] + K, JH))
    cross = []
    orig = mdl._cross_reconstruct

    def rec(*a, **k):
        out = orig(*a, **k)
        cross.append((N(out[0]), N(out[1])))
        return out

    mdl._cross_reconstruct = rec
    with E.use_rng(E.ReplayRng(s)):
        mdl.image_idbn.train(1)
        for i, r in enumerate(mdl.image_idbn.layers):
            check_state(r, fx, f"img{i}_", 1e-4)
        mdl.train_joint(m["joint_epochs"])
        del mdl._cross_reconstruct
        check_state(mdl.joint_rbm, fx, "joint_", rel)
        assert_close(N(mdl.z_class_mean), fx["z_class_mean"], 5e-5, "z_class_mean")
        cd = torch.cat([h["cd_losses"] for h in mdl.joint_history if h["cd_losses"] is not None]).numpy()
        assert_close(cd, fx["cd_losses"], 2e-4, "cd losses")
        for e in (0, 7, 8, 9):
            assert_close(cross[e * NB + NB - 1][1], fx[f"cross_py_e{e}_last"], rel, f"p_y epoch {e}")
            assert_close(cross[e * NB + NB - 1][0], fx[f"cross_img_e{e}_last"], rel, f"img epoch {e}")
        py = np.array([sum(float(c[1].astype(np.float64).sum()) for c in cross[e * NB:(e + 1) * NB]) for e in range(10)])
        im = np.array([sum(float(c[0].astype(np.float64).sum()) for c in cross[e * NB:(e + 1) * NB]) for e in range(10)])
        assert_close(py, fx["cross_py_sum_per_epoch"], 5e-5, "sum p_y per epoch")
        assert_close(im, fx["cross_img_sum_per_epoch"], 2e-4, "sum img per epoch")
        assert_close(N(mdl.represent((torch.from_numpy(X[:8]), torch.from_numpy(Y[:8])))), fx["represent"], 2e-4,
                     "iMDBN.represent")
        zi = mdl.image_idbn.represent(torch.from_numpy(X[:8]))
        a, b = mdl._cross_reconstruct(zi, T(Y[:8], dev), steps=9)
        assert_close(N(a), fx["xr_img"], rel, "xr img"); assert_close(N(b), fx["xr_py"], rel, "xr p_y")
        zcm = mdl.z_class_mean
        mdl.z_class_mean = None
        a, b = mdl._cross_reconstruct(zi, T(Y[:8], dev))
        assert_close(N(a), fx["xr_nomu_img"], rel, "xr nomu img"); assert_close(N(b), fx["xr_nomu_py"], rel, "xr nomu p_y")
        mdl.z_class_mean = zcm
    assert s.exhausted_cat()
    # online metrics (imdbn.py:615-657) against the values the reference itself computed, epoch by epoch
    assert len(mdl.joint_history) == m["joint_epochs"]
    got = np.array([[h["text_top1"], h["text_top3"], h["text_ce"], h["image_mse"]] for h in mdl.joint_history])
    want = fx["joint_metrics"]
    n = B * NB
    assert np.abs(got[:, :2] - want[:, :2]).max() <= 1.0 / n + 1e-12, "top-1 / top-3 differ by more than one sample"
    assert_close(got[:, 2], want[:, 2], 2e-4, "text CE per epoch")
    assert_close(got[:, 3], want[:, 3], 2e-4, "image MSE per epoch")
    return mdl


def case_pretrained_finetune(dev, rel=1e-4):
    """a17 (imdbn.py:294-384): load_pretrained_image_idbn on the reference-written pickle re-binds the layer tensors and
    re-zeros the momentum buffers; finetune_image_last_layer trains the last layer at lr x lr_scale on the lower layers'
    representation and restores lr.  Fixture generated by the unmodified reference."""
    import os
    fx = Fixture("pretrained_finetune_100_40_20.npz")
    m = fx.meta
    s = fx.stream()
    sizes, JH, K, B, NB = m["sizes"], m["joint_hidden"], m["K"], m["B"], m["NB"]
    X = (s.uniform((B * NB, 100)) > 0.75).astype(F32)
    Y = np.eye(K, dtype=F32)[fx["yi"]]
    dl = loader(X, Y, B)
    mdl = iMDBN(sizes, JH, params=dict(m["params"]), dataloader=dl, val_loader=dl, device=torch.device(dev), num_labels=K)
    before = [r.W.data_ptr() for r in mdl.image_idbn.layers]
    for r in mdl.image_idbn.layers:                       # make the momentum buffers non-zero: loading must re-zero them
        r.W_m = torch.ones_like(r.W.data)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_idbn_small.pkl")
    assert mdl.load_pretrained_image_idbn(path) is True
    assert not mdl.load_pretrained_image_idbn(path + ".missing")
    for i, r in enumerate(mdl.image_idbn.layers):
        assert r.W.data_ptr() != before[i] and r.W.device.type == torch.device(dev).type          # re-bound, on the device
        assert float(r.W_m.abs().sum()) == 0.0 and float(r.hb_m.abs().sum()) == 0.0 and float(r.vb_m.abs().sum()) == 0.0
        assert abs(float(r.W.detach().double().sum()) - float(fx[f"loaded{i}_W_sum"])) < 1e-6
    last = mdl.image_idbn.layers[-1]
    lr0 = float(last.lr)
    assert abs(lr0 - m["lr0"]) < 1e-12
    mdl.finetune_image_last_layer(epochs=0)               # no-op (imdbn.py:357-358)
    with E.use_rng(E.ReplayRng(s)):
        mdl.finetune_image_last_layer(epochs=m["epochs"], lr_scale=m["lr_scale"])
    assert float(last.lr) == lr0                          # restored (imdbn.py:383)
    losses = torch.cat(mdl.finetune_losses).numpy()
    assert_close(losses, fx["losses"], 5e-5, "fine-tuning losses")
    check_state(last, fx, "last_", rel)
    check_state(mdl.image_idbn.layers[0], fx, "first_", 1e-7)
    return mdl


def case_live_best_of_k(dev, V=150, H=48, Dz=140, B=9, K=7, rel=1e-4):
    """Opt-in live best-of-K: free energies match the oracle, every row keeps its
    lowest-energy candidate, the default (reference-identical, inert) path is untouched by the flag being off."""
    import oracle.rbm_oracle as O
    g = np.random.default_rng(5)
    X = (g.random((B * 2, 100)) > 0.8).astype(F32)
    Y = np.eye(V - Dz, dtype=F32)[np.arange(B * 2) % (V - Dz)]
    dl = loader(X, Y, B)
    params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95,
              "LEARNING_RATE_DYNAMIC": True, "CD": 1, "CROSS_GIBBS_STEPS": 6, "CROSS_LIVE_BEST_OF_K": True, "CROSS_BEST_OF_K": K}
    mdl = iMDBN([100, Dz], H, params=params, dataloader=dl, val_loader=dl, device=torch.device(dev), num_labels=V - Dz)
    from oracle.draws import DrawStream
    s = DrawStream(7)
    W0 = (s.normal((V, H)) / F32(np.sqrt(V))).astype(F32)
    hb = (s.normal((H,)) * F32(0.1)).astype(F32); vb = (s.normal((V,)) * F32(0.1)).astype(F32)
    set_params(mdl.joint_rbm, dev, W0, hb, vb)
    st = O.RBMState.create(W0, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, softmax_groups=[(Dz, V)], hid_bias=hb, vis_bias=vb)
    assert mdl.live_best_of_k and mdl.best_of_k == K
    z = g.random((B, Dz), dtype=F32)
    y = Y[:B]
    vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
    vk[:, Dz:] = y; km[:, Dz:] = 1
    with E.use_rng(E.ReplayRng(DrawStream(3))):
        v0 = mdl.joint_rbm.noisy_meanfield_annealed(T(vk, dev), T(km, dev), n_steps=5)
        pick, cands, en = mdl._best_of_k(v0, T(km, dev), K)
    cands, en, pick = N(cands), N(en), N(pick)
    assert cands.shape == (K, B, V) and en.shape == (K, B)
    for k in range(K):
        assert_close(en[k], O.free_energy(st, cands[k]), rel, f"free energy of candidate {k}")
    best = en.argmin(0)
    np.testing.assert_array_equal(pick, cands[best, np.arange(B)])
    # the refinement is a deterministic mean-field step: candidate k+1 = one clamped down(up(.)) pass at T = 0.9
    return en


def case_bimodal_small(dev, rel=3e-4):
    """iMDBN_BiModal through the product classes against the fixture generated from the reference class
    (warm-up epochs 0-7: clamped CD-3 with sampled hidden units; epochs 8-9: two-layer joint CD + clamped CD-3;
    bidirectional Gibbs cross reconstruction on every batch)."""
    fx = Fixture("bimodal_small_100_40_20__64_30_16__j24_12.npz")
    m = fx.meta
    s = fx.stream()
    K, B, NB = m["K"], m["B"], m["NB"]
    Nn = B * NB
    yi = fx["yi"]
    X1 = np.abs((s.uniform((K, 100)) > 0.7).astype(F32)[yi] - (s.uniform((Nn, 100)) > 0.9).astype(F32)).astype(F32)
    X2 = np.abs((s.uniform((K, 64)) > 0.6).astype(F32)[yi] - (s.uniform((Nn, 64)) > 0.92).astype(F32)).astype(F32)
    dl = loader(X1, X2, B)
    mdl = iMDBN_BiModal(m["sizes1"], m["sizes2"], m["joint"], params=dict(m["params"]), dataloader=dl, val_loader=dl,
                        device=torch.device(dev))
    for sizes, dbn in ((m["sizes1"], mdl.mod1_dbn), (m["sizes2"], mdl.mod2_dbn)):
        for i, r in enumerate(dbn.layers):
            set_params(r, dev, init_W(s, sizes[i], sizes[i + 1]))
    vis = m["sizes1"][-1] + m["sizes2"][-1]
    for r, h in zip(mdl.joint_layers, m["joint"]):
        set_params(r, dev, init_W(s, vis, h))
        vis = h
    cross = []
    orig = mdl._cross_reconstruct

    def rec(*a, **k):
        out = orig(*a, **k)
        cross.append((N(out[0]), N(out[1])))
        return out

    mdl._cross_reconstruct = rec
    with E.use_rng(E.ReplayRng(s)):
        mdl.train_joint(m["joint_epochs"])
        del mdl._cross_reconstruct
        for li, r in enumerate(mdl.joint_layers):
            check_state(r, fx, f"joint{li}_", rel)
        cd = torch.cat([h["cd_losses"] for h in mdl.joint_history if h["cd_losses"] is not None]).numpy()
        assert_close(cd, fx["cd_losses"], 2e-4, "cd losses")
        for e in (0, 7, 8, 9):
            assert_close(cross[e * NB + NB - 1][0], fx[f"cross_m1_e{e}_last"], rel, f"mod1<-mod2 epoch {e}")
            assert_close(cross[e * NB + NB - 1][1], fx[f"cross_m2_e{e}_last"], rel, f"mod2<-mod1 epoch {e}")
        s1 = np.array([sum(float(c[0].astype(np.float64).sum()) for c in cross[e * NB:(e + 1) * NB]) for e in range(10)])
        s2 = np.array([sum(float(c[1].astype(np.float64).sum()) for c in cross[e * NB:(e + 1) * NB]) for e in range(10)])
        assert_close(s1, fx["cross_m1_sum_per_epoch"], 2e-4, "sum mod1 per epoch")
        assert_close(s2, fx["cross_m2_sum_per_epoch"], 2e-4, "sum mod2 per epoch")
        assert_close(N(mdl.represent((torch.from_numpy(X1[:8]), torch.from_numpy(X2[:8])))), fx["represent"], 2e-4,
                     "iMDBN_BiModal.represent")
        z1 = mdl.mod1_dbn.represent(torch.from_numpy(X1[:8]))
        z2 = mdl.mod2_dbn.represent(torch.from_numpy(X2[:8]))
        a, b = mdl._cross_reconstruct(z1, z2, steps=9)
        assert_close(N(a), fx["xr_m1"], rel, "xr mod1"); assert_close(N(b), fx["xr_m2"], rel, "xr mod2")
    # the online metric of the last epoch is the mean squared error over the epoch's batches
    h = mdl.joint_history[-1]
    assert h["mod1_mse"] is not None and 0.0 < h["mod1_mse"] < 1.0 and 0.0 < h["mod2_mse"] < 1.0
    return mdl


def case_class_free_energies(dev, V=150, Dz=140, H=48, B=21, rel=2e-5):
    """imdbn.utils.energy_utils against the reference formula (energy_utils.py:31-56) evaluated with the oracle."""
    import oracle.rbm_oracle as O
    from oracle.draws import DrawStream
    from imdbn.utils import class_free_energies, rbm_free_energy
    K = V - Dz
    s = DrawStream(11)
    W0 = (s.normal((V, H)) / F32(np.sqrt(V))).astype(F32)
    hb = (s.normal((H,)) * F32(0.1)).astype(F32); vb = (s.normal((V,)) * F32(0.1)).astype(F32)
    r = RBM(V, H, 0.1, 1e-4, 0.5, softmax_groups=[(Dz, V)])
    set_params(r, dev, W0, hb, vb)
    st = O.RBMState.create(W0, 0.1, 1e-4, 0.5, softmax_groups=[(Dz, V)], hid_bias=hb, vis_bias=vb)
    z = s.uniform((B, Dz)).astype(F32)
    got = N(class_free_energies(r, T(z, dev), K, Dz))
    want = np.stack([O.free_energy(st, np.concatenate([z, np.tile(np.eye(K, dtype=F32)[k], (B, 1))], 1)) for k in range(K)], 1)
    assert got.shape == (B, K)
    assert_close(got, want, rel, "class free energies")
    v = np.concatenate([z, np.eye(K, dtype=F32)[np.arange(B) % K]], 1)
    assert_close(N(rbm_free_energy(r, T(v, dev))), O.free_energy(st, v), rel, "rbm_free_energy")
