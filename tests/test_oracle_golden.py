"""Pin the CPU oracle (oracle/rbm_oracle.py) against golden vectors produced by the
unmodified reference (tests/golden/make_fixtures.py).  CPU-only; no product code involved.

Tolerances: the oracle and the reference differ only in GEMM summation order and in the
last ulp of exp(), so continuous outputs agree to ~1e-6 relative; sampled paths agree
exactly unless a Bernoulli margin |p-u| is below that noise (the fixtures record their
minimum margin).
"""
import os

import numpy as np
import pytest

import oracle.rbm_oracle as O
from golden_utils import Fixture, assert_close, init_W

F32 = np.float32
PARAM_KEYS = ("W", "hid_bias", "vis_bias", "W_m", "hb_m", "vb_m")


def _check_state(st, fx, prefix, rel=2e-5):
    for k in PARAM_KEYS:
        if prefix + k in fx.a:
            assert_close(getattr(st, k), fx[prefix + k], rel=rel, what=prefix + k, atol=2e-6)


def test_c1_rbm784x256_cd1_20_updates():
    fx = Fixture("c1_rbm784x256_cd1.npz")
    m = fx.meta
    s = fx.stream()
    st = O.RBMState.create(init_W(s, m["V"], m["H"]), m["lr"], m["wd"], m["mom"], dynamic_lr=True,
                           final_momentum=m["final_momentum"])
    X = (s.uniform((640, m["V"])) > 0.5).astype(F32)
    losses = []
    for i in range(m["updates"]):
        losses.append(O.train_epoch(st, X[32 * i:32 * i + 32], m["epoch"], m["CD"], s))
        if i == 0:
            assert_close(st.W[::16], fx["a1_W_rows"], rel=1e-6, what="W after 1 update")
            assert_close(st.W_m[::16], fx["a1_W_m_rows"], rel=1e-5, what="W_m after 1 update")
            assert_close(st.hid_bias, fx["a1_hid_bias"], rel=1e-5, what="hid_bias after 1 update")
            assert_close(st.vis_bias, fx["a1_vis_bias"], rel=1e-5, what="vis_bias after 1 update")
    assert_close(np.array(losses, F32), fx["losses"], rel=1e-6, what="losses")
    assert_close(st.W, fx["W"], rel=1e-5, what="W after 20 updates")
    for k in ("hid_bias", "vis_bias", "hb_m", "vb_m"):
        assert_close(getattr(st, k), fx[k], rel=2e-5, what=k)
    assert_close(st.W_m[::16], fx["W_m_rows"], rel=2e-5, what="W_m rows")


def test_joint_small_every_rbm_method():
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
        return O.RBMState.create(W0, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95,
                                 softmax_groups=[(96, 104)], hid_bias=hb0, vis_bias=vb0, **kw)

    def clamp(which):
        vk = np.zeros((B, V), F32); km = np.zeros((B, V), F32)
        if which == "y":
            vk[:, Dz:] = y; km[:, Dz:] = 1
        else:
            vk[:, :Dz] = z; km[:, :Dz] = 1
        return vk, km

    r = fresh()
    assert_close(O.forward(r, data_real), fx["fwd_T1"], 1e-6, "forward T=1")
    assert_close(O.forward(r, data_real, T=2.5), fx["fwd_T25"], 1e-6, "forward T=2.5")
    th = O.forward(r, data_real)
    assert_close(O.visible_probs(r, th), fx["vis_T1"], 1e-6, "visible_probs")
    assert_close(O.visible_probs(r, th, T=0.7), fx["vis_T07"], 1e-6, "visible_probs T=.7")
    assert_close(O.backward(r, th, return_logits=True), fx["bwd_logits"], 1e-6, "backward logits")
    assert_close(O.backward(r, th), fx["bwd"], 1e-6, "backward")
    np.testing.assert_array_equal(O.sample_visible(r, O.visible_probs(r, th), s), fx["sample_visible"])
    np.testing.assert_array_equal(O.backward_sample(r, th, s), fx["backward_sample"])
    g = O.gibbs_step(r, data, s)
    np.testing.assert_array_equal(g[0], fx["gibbs_v_next"])
    assert_close(g[1], fx["gibbs_v_prob"], 1e-6, "gibbs v_prob")
    np.testing.assert_array_equal(g[2], fx["gibbs_h"])
    assert_close(g[3], fx["gibbs_h_prob"], 1e-6, "gibbs h_prob")

    r = fresh(sparsity=True, sparsity_factor=0.1)
    loss = O.train_epoch(r, data, 7, 2, s)
    assert_close(loss, fx["te_loss"], 1e-6, "train_epoch CD2 loss")
    _check_state(r, fx, "te_")

    r = fresh()
    l1 = O.train_epoch(r, data_real, 0, 1, s)
    l2 = O.train_epoch(r, data_real, 1, 1, s)
    assert_close(np.array([l1, l2], F32), fx["ter_loss"], 1e-6, "train_epoch real loss")
    _check_state(r, fx, "ter_")

    r = fresh()
    vk, km = clamp("z")
    assert_close(O.conditional_gibbs(r, vk, km, s, n_steps=10), fx["cg_plain"], 1e-5, "conditional_gibbs")
    assert_close(O.conditional_gibbs(r, vk, km, s, n_steps=5, sample_h=True, sample_v=True), fx["cg_sampled"],
                 1e-5, "conditional_gibbs sampled")
    assert_close(O.conditional_gibbs_annealed(r, vk, km, s, n_steps=12, T0=2.5, T1=1.0, sample_h_until=6,
                                              sample_v_every=2, final_meanfield=True), fx["cga"], 1e-5, "cga")
    assert_close(O.conditional_gibbs_annealed(r, vk, km, s, n_steps=6, sample_h_until=0, final_meanfield=False),
                 fx["cga_nofinal"], 1e-5, "cga no final")
    vk, km = clamp("y")
    assert_close(O.noisy_meanfield_annealed(r, vk, km, s, n_steps=20), fx["nmf20"], 1e-5, "nmf20")
    r.mu_pull = {"mu_k": mu, "eta0": 0.15}
    got = O.noisy_meanfield_annealed(r, vk, km, s, n_steps=20, sharpen_last=3)
    assert_close(got, fx["nmf20_mu"], 1e-5, "nmf20 mu-pull")
    assert_close(O.noisy_meanfield_annealed(r, fx["nmf20_mu"], km, s, n_steps=1, T0=0.9, T1=0.9, sigma0=0.0,
                                            hot_frac=0.0, sharpen_last=0, T_cold_plus=0.9), fx["nmf1_mu"], 1e-5,
                 "nmf 1-step refinement")
    r.mu_pull = None

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
        loss = O.train_epoch_clamped(r, vk, km, 9, s, **kw)
        assert_close(loss, fx[tag + "_loss"], 2e-5, tag + " loss")
        _check_state(r, fx, tag + "_", rel=5e-5)
    assert s.exhausted_cat()
    got_log = ";".join(f"{k}{'x'.join(map(str, sh))}" for k, sh in s.log)
    assert got_log == m["draw_log"], "draw order differs from the reference's (Appendix B)"


def test_idbn_small_stack():
    fx = Fixture("idbn_small_100_40_20.npz")
    m = fx.meta
    s = fx.stream()
    sizes, N, B = m["sizes"], m["N"], m["B"]
    X = (s.uniform((N, sizes[0])) > 0.8).astype(F32)
    p = m["params"]
    layers = []
    for i in range(len(sizes) - 1):
        layers.append(O.RBMState.create(
            init_W(s, sizes[i], sizes[i + 1]), p["LEARNING_RATE"], p["WEIGHT_PENALTY"], p["INIT_MOMENTUM"],
            dynamic_lr=p["LEARNING_RATE_DYNAMIC"], final_momentum=p["FINAL_MOMENTUM"],
            sparsity=(p["SPARSITY"] and i == len(sizes) - 2), sparsity_factor=p["SPARSITY_FACTOR"]))
    losses = []
    for epoch in range(m["epochs"]):
        for b in range(N // B):
            losses += O.idbn_train_batch(layers, X[b * B:(b + 1) * B], epoch, p["CD"], s)
    assert_close(np.array(losses, F32), fx["losses"], 1e-5, "losses")
    for i, st in enumerate(layers):
        _check_state(st, fx, f"L{i}_", rel=5e-5)
    assert_close(O.idbn_represent(layers, X[:8]), fx["represent"], 1e-5, "represent")
    assert_close(O.idbn_represent(layers, X[:8], upto_layer=1), fx["represent_l1"], 1e-5, "represent l1")
    assert_close(O.idbn_reconstruct(layers, X[:8]), fx["reconstruct"], 1e-5, "reconstruct")
    assert_close(O.idbn_decode(layers, O.idbn_represent(layers, X[:8])), fx["decode"], 1e-5, "decode")
    assert len(s.log) == m["draw_log_len"]


def test_imdbn_small_train_joint_and_cross_reconstruct():
    fx = Fixture("imdbn_small_100_40_20_j16.npz")
    m = fx.meta
    s = fx.stream()
    sizes, JH, K, B, NB = m["sizes"], m["joint_hidden"], m["K"], m["B"], m["NB"]
    N = B * NB
    yi = fx["yi"]
    proto = (s.uniform((K, 100)) > 0.7).astype(F32)
    flip = (s.uniform((N, 100)) > 0.9).astype(F32)
    X = np.abs(proto[yi] - flip).astype(F32)
    Y = np.eye(K, dtype=F32)[yi]
    p = m["params"]
    layers = []
    for i in range(len(sizes) - 1):
        layers.append(O.RBMState.create(
            init_W(s, sizes[i], sizes[i + 1]), p["LEARNING_RATE"], p["WEIGHT_PENALTY"], p["INIT_MOMENTUM"],
            dynamic_lr=True, final_momentum=p["FINAL_MOMENTUM"],
            sparsity=(p["SPARSITY"] and i == len(sizes) - 2), sparsity_factor=p["SPARSITY_FACTOR"]))
    joint = O.RBMState.create(init_W(s, sizes[-1] + K, JH), p["JOINT_LEARNING_RATE"], p["WEIGHT_PENALTY"],
                              p["INIT_MOMENTUM"], dynamic_lr=True, final_momentum=p["FINAL_MOMENTUM"],
                              softmax_groups=[(sizes[-1], sizes[-1] + K)])
    batches = [(X[b * B:(b + 1) * B], Y[b * B:(b + 1) * B]) for b in range(NB)]
    for img, _ in batches:                                   # image_idbn.train(1)
        O.idbn_train_batch(layers, img, 0, p["CD"], s)
    for i, st in enumerate(layers):
        _check_state(st, fx, f"img{i}_", rel=5e-5)
    zcm = O.init_joint_bias_from_data(layers, joint, batches, K, n_batches=10)       # train_joint prologue
    cd_losses, py_sums, img_sums, metrics = [], [], [], []
    for epoch in range(m["joint_epochs"]):
        ps = 0.0; isum = 0.0
        tot = np.zeros(5, np.float64)                       # n, top1, top3, ce_sum, mse_sum  (imdbn.py:544-551)
        for b_idx, (img, y) in enumerate(batches):
            r = O.train_joint_batch(layers, joint, img, y, epoch, b_idx, s, p["JOINT_CD"],
                                    p["JOINT_AUX_COND_STEPS"], p["CROSS_GIBBS_STEPS"], z_class_mean=zcm)
            tot += np.array([r["n"], r["top1"], r["top3"], r["ce_sum"], r["mse_sum"]], np.float64)
            if r["loss_cd"] is not None:
                cd_losses.append(r["loss_cd"])
            ps += float(r["p_y"].astype(np.float64).sum()); isum += float(r["img_from_txt"].astype(np.float64).sum())
            if b_idx == NB - 1 and epoch in (0, 7, 8, 9):
                assert_close(r["p_y"], fx[f"cross_py_e{epoch}_last"], 2e-4, f"p_y epoch {epoch}")
                assert_close(r["img_from_txt"], fx[f"cross_img_e{epoch}_last"], 2e-4, f"img epoch {epoch}")
        py_sums.append(ps); img_sums.append(isum)
        metrics.append([tot[1] / tot[0], tot[2] / tot[0], tot[3] / tot[0], tot[4] / (tot[0] * img.shape[1])])   # :648-657
    # the reference's online metrics (recorded from its own F.binary_cross_entropy / F.mse_loss calls)
    metrics = np.array(metrics)
    np.testing.assert_allclose(metrics[:, :2], fx["joint_metrics"][:, :2], rtol=0, atol=1e-12, err_msg="top-1 / top-3")
    assert_close(metrics[:, 2], fx["joint_metrics"][:, 2], 1e-5, "text CE per epoch")
    assert_close(metrics[:, 3], fx["joint_metrics"][:, 3], 1e-5, "image MSE per epoch")
    assert_close(zcm, fx["z_class_mean"], 1e-5, "z_class_mean")
    assert_close(np.array(cd_losses, F32), fx["cd_losses"], 1e-4, "cd losses")
    assert_close(np.array(py_sums), fx["cross_py_sum_per_epoch"], 1e-5, "sum p_y per epoch")
    assert_close(np.array(img_sums), fx["cross_img_sum_per_epoch"], 1e-4, "sum img per epoch")
    _check_state(joint, fx, "joint_", rel=2e-4)
    assert_close(O.joint_represent(layers, joint, X[:8], Y[:8]), fx["represent"], 1e-4, "iMDBN.represent")
    zi = O.idbn_represent(layers, X[:8])
    a, b = O.cross_reconstruct(layers, joint, zi, Y[:8], 9, s, z_class_mean=zcm)
    assert_close(a, fx["xr_img"], 2e-4, "xr img"); assert_close(b, fx["xr_py"], 2e-4, "xr p_y")
    a, b = O.cross_reconstruct(layers, joint, zi, Y[:8], p["CROSS_GIBBS_STEPS"], s, z_class_mean=None)
    assert_close(a, fx["xr_nomu_img"], 2e-4, "xr nomu img"); assert_close(b, fx["xr_nomu_py"], 2e-4, "xr nomu p_y")
    assert s.exhausted_cat()


def _bimodal_inputs(fx):
    m = fx.meta
    s = fx.stream()
    K, B, NB = m["K"], m["B"], m["NB"]
    N = B * NB
    yi = fx["yi"]
    X1 = np.abs((s.uniform((K, 100)) > 0.7).astype(F32)[yi] - (s.uniform((N, 100)) > 0.9).astype(F32)).astype(F32)
    X2 = np.abs((s.uniform((K, 64)) > 0.6).astype(F32)[yi] - (s.uniform((N, 64)) > 0.92).astype(F32)).astype(F32)
    return s, X1, X2


def test_bimodal_small_train_joint_and_cross_reconstruct():
    """iMDBN_BiModal (imdbn_bimodal.py:617-829): warm-up + main phase on a two-layer joint DBN."""
    fx = Fixture("bimodal_small_100_40_20__64_30_16__j24_12.npz")
    m = fx.meta
    s, X1, X2 = _bimodal_inputs(fx)
    p = m["params"]
    B, NB = m["B"], m["NB"]

    def stack(sizes):
        return [O.RBMState.create(init_W(s, sizes[i], sizes[i + 1]), p["LEARNING_RATE"], p["WEIGHT_PENALTY"], p["INIT_MOMENTUM"],
                                  dynamic_lr=True, final_momentum=p["FINAL_MOMENTUM"],
                                  sparsity=(p["SPARSITY"] and i == len(sizes) - 2), sparsity_factor=p["SPARSITY_FACTOR"])
                for i in range(len(sizes) - 1)]

    l1, l2 = stack(m["sizes1"]), stack(m["sizes2"])
    joint, vis = [], m["sizes1"][-1] + m["sizes2"][-1]
    for h in m["joint"]:
        joint.append(O.RBMState.create(init_W(s, vis, h), p["JOINT_LEARNING_RATE"], p["WEIGHT_PENALTY"], p["INIT_MOMENTUM"],
                                       dynamic_lr=True, final_momentum=p["FINAL_MOMENTUM"]))
        vis = h
    batches = [(X1[b * B:(b + 1) * B], X2[b * B:(b + 1) * B]) for b in range(NB)]
    O.bimodal_init_joint_bias(l1, l2, joint[0], batches, n_batches=10)
    cd_losses, s1, s2 = [], [], []
    for epoch in range(m["joint_epochs"]):
        a1 = a2 = 0.0
        for b_idx, (m1, m2) in enumerate(batches):
            r = O.bimodal_train_joint_batch(l1, l2, joint, m1, m2, epoch, s, p["JOINT_CD"], p["JOINT_AUX_COND_STEPS"],
                                            p["CROSS_GIBBS_STEPS"])
            if r["loss_cd"] is not None:
                cd_losses.append(r["loss_cd"])
            a1 += float(r["mod1_from_mod2"].astype(np.float64).sum()); a2 += float(r["mod2_from_mod1"].astype(np.float64).sum())
            if b_idx == NB - 1 and epoch in (0, 7, 8, 9):
                assert_close(r["mod1_from_mod2"], fx[f"cross_m1_e{epoch}_last"], 2e-4, f"mod1<-mod2 epoch {epoch}")
                assert_close(r["mod2_from_mod1"], fx[f"cross_m2_e{epoch}_last"], 2e-4, f"mod2<-mod1 epoch {epoch}")
        s1.append(a1); s2.append(a2)
    assert_close(np.array(cd_losses, F32), fx["cd_losses"], 1e-4, "cd losses")
    assert_close(np.array(s1), fx["cross_m1_sum_per_epoch"], 1e-4, "sum mod1 per epoch")
    assert_close(np.array(s2), fx["cross_m2_sum_per_epoch"], 1e-4, "sum mod2 per epoch")
    for li, st in enumerate(joint):
        _check_state(st, fx, f"joint{li}_", rel=2e-4)
    assert_close(O.bimodal_represent(l1, l2, joint, X1[:8], X2[:8]), fx["represent"], 1e-4, "iMDBN_BiModal.represent")
    a, b = O.bimodal_cross_reconstruct(l1, l2, joint[0], O.idbn_represent(l1, X1[:8]), O.idbn_represent(l2, X2[:8]), 9, s)
    assert_close(a, fx["xr_m1"], 2e-4, "xr mod1"); assert_close(b, fx["xr_m2"], 2e-4, "xr mod2")


@pytest.mark.slow
def test_c2_headline_digest():
    """10000<->1500, batch 64, 3 updates: digests only (the weights are 60 MB)."""
    fx = Fixture("c2_rbm10000x1500_cd1_digest.npz")
    m = fx.meta
    s = fx.stream()
    V, H, B, U = m["V"], m["H"], m["B"], m["updates"]
    st = O.RBMState.create(init_W(s, V, H), 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95)
    X = (s.uniform((B * U, V)) > 0.9).astype(F32)
    losses = [O.train_epoch(st, X[B * i:B * i + B], 0, 1, s) for i in range(U)]
    assert_close(np.array(losses, F32), fx["losses"], 1e-5, "losses")
    for k in ("W", "W_m"):
        a = getattr(st, k)
        assert abs(a.astype(np.float64).sum() - fx[k + "_sum"]) <= 1e-4 * abs(fx[k + "_sum"]) + 1e-3
        assert abs((a.astype(np.float64) ** 2).sum() - fx[k + "_sumsq"]) <= 1e-4 * fx[k + "_sumsq"]
        assert_close(a.ravel()[fx[k + "_probe_idx"]], fx[k + "_probe_val"], 1e-4, k + " probes", atol=1e-6)
    for k in ("hid_bias", "vis_bias", "hb_m", "vb_m"):
        assert_close(getattr(st, k), fx[k], 1e-4, k, atol=1e-6)


def test_pretrained_idbn_is_loaded_and_its_last_layer_fine_tuned():
    """imdbn.py:294-384: the reference pickle's layers (dict form), momentum re-zeroed, 2 fine-tuning epochs of the last
    layer at lr x 0.3 on the representation of the lower layers; the oracle replays the fixture's draws."""
    import pickle
    import imdbn.models  # noqa: F401  (the classes the reference pickle names)
    fx = Fixture("pretrained_finetune_100_40_20.npz")
    m = fx.meta
    s = fx.stream()
    B, NB = m["B"], m["NB"]
    X = (s.uniform((B * NB, 100)) > 0.75).astype(F32)
    with open(os.path.join(os.path.dirname(__file__), "golden", "ref_idbn_small.pkl"), "rb") as f:
        obj = pickle.load(f)
    p = m["params"]
    layers = []
    for i, r in enumerate(obj["layers"]):
        st = O.RBMState.create(r.W.detach().cpu().numpy().astype(F32), p["LEARNING_RATE"], p["WEIGHT_PENALTY"], p["INIT_MOMENTUM"],
                               dynamic_lr=True, final_momentum=p["FINAL_MOMENTUM"],
                               sparsity=bool(r.sparsity), sparsity_factor=float(r.sparsity_factor),
                               hid_bias=r.hid_bias.detach().cpu().numpy().astype(F32), vis_bias=r.vis_bias.detach().cpu().numpy().astype(F32))
        assert abs(float(st.W.astype(np.float64).sum()) - float(fx[f"loaded{i}_W_sum"])) < 1e-9
        layers.append(st)                                   # momentum buffers start at zero (imdbn.py:329-331)
    last = layers[-1]
    last.lr = max(1e-8, m["lr0"] * m["lr_scale"])           # :363
    losses = []
    for ep in range(m["epochs"]):
        for b in range(NB):
            v = X[b * B:(b + 1) * B]
            for st in layers[:-1]:
                v = O.forward(st, v)
            losses.append(O.train_epoch(last, v, ep, p["CD"], s))
    assert_close(np.array(losses, F32), fx["losses"], 1e-5, "fine-tuning losses")
    _check_state(last, fx, "last_", rel=5e-5)
    _check_state(layers[0], fx, "first_", rel=0.0)
