"""CPU: the oracle (oracle/diffusion_oracle.py) replayed against every golden fixture the
reference produced (tests/golden/make_goldens.py).  Tolerances are fp32 round-off only: the
oracle runs the same torch CPU kernels as the reference did, so most comparisons are exact."""
import numpy as np
import pytest
import torch

from oracle import diffusion_oracle as O

SM = dict(mutation_dim=8, expression_dim=24, pathway_dim=8, condition_dim=3)
SM_H = [32, 64, 32]
NH, TD = 3, 128


def load(golden_dir, name):
    return dict(np.load(golden_dir / f"{name}.npz"))


def sd_from(g, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def params_only(sd):
    return {k: v for k, v in sd.items() if k.startswith(("condition_embed", "unet"))}


def close(a, b, rtol=0.0, atol=0.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    tol = atol + rtol * np.abs(b).max()
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= tol, f"max|d|={np.abs(a - b).max():.3e} tol={tol:.3e}"


def test_g1_schedule_bit_exact(golden_dir):
    g = load(golden_dir, "g1_schedule")
    for sched in ("cosine", "linear"):
        for T in (1000, 50):
            bufs = O.schedule_buffers(sched, T)
            for k, v in bufs.items():
                assert np.array_equal(v.numpy(), g[f"{sched}_{T}_{k}"]), (sched, T, k)
    with pytest.raises(ValueError):
        O.beta_schedule("sigmoid", 10)


def test_g2_time_embedding_bit_exact(golden_dir):
    g = load(golden_dir, "g2_time_embedding")
    emb = O.time_embedding(torch.from_numpy(g["t_norm_train"]), 128)
    assert np.array_equal(emb.numpy(), g["emb"])
    # train-path t.float()/T and sample-path python t/T agree bit for bit
    assert np.array_equal(g["t_norm_train"], g["t_norm_sample"])


def test_g3_eval_forward_and_taps(golden_dir):
    g = load(golden_dir, "g3g4_small_model")
    sd = sd_from(g)
    x, cond, t = (torch.from_numpy(g[k]) for k in ("x", "cond", "t"))
    c_emb = O.condition_embed(sd, cond)
    close(c_emb, g["eval_c_emb"], rtol=1e-6)
    taps = {}
    pred = O.unet_forward(sd, x, t.float() / 1000, c_emb, NH, TD, taps=taps)
    close(pred, g["eval_noise_pred"], rtol=2e-6)
    for name in O.block_names(NH):
        close(taps[name], g["eval_tap." + name[len("unet."):]], rtol=2e-6)
    close(taps["t_emb"], g["eval_tap.time_proj"], rtol=1e-6)
    close(taps["c_proj"], g["eval_tap.cond_proj"], rtol=1e-6)


def test_g4_q_sample_loss_grads(golden_dir):
    g = load(golden_dir, "g3g4_small_model")
    sd = sd_from(g)
    bufs = O.schedule_buffers("cosine", 1000)
    x, cond, t, noise = (torch.from_numpy(g[k]) for k in ("x", "cond", "t", "noise"))
    assert np.array_equal(O.q_sample(bufs, x, t, noise).numpy(), g["q_sample_x_t"])
    p = params_only(sd)
    loss, grads = O.training_loss_and_grads(p, bufs, x, cond, t, noise, NH, TD)
    close(loss, g["eval_loss"], rtol=1e-6)
    for k, gr in grads.items():
        close(gr, g["eval_grad." + k], rtol=2e-5, atol=1e-9)
    pred = O.training_forward(p, bufs, x, cond, t, noise, NH, TD, return_loss=False)
    close(pred, g["eval_forward_noise_pred"], rtol=2e-6)
    # train mode with the recorded dropout keep-masks (p = 0.2)
    masks = [torch.from_numpy(g[f"train_mask.{i}"]) for i in range(5)]
    loss, grads = O.training_loss_and_grads(p, bufs, x, cond, t, noise, NH, TD, masks, 0.2)
    close(loss, g["train_loss"], rtol=1e-6)
    for k, gr in grads.items():
        close(gr, g["train_grad." + k], rtol=2e-5, atol=1e-9)


def _small_sd(golden_dir):
    return params_only(sd_from(load(golden_dir, "g3g4_small_model")))


def test_g5_p_sample_steps_and_chains(golden_dir):
    g = load(golden_dir, "g5_sampling")
    sd = _small_sd(golden_dir)
    cond = torch.from_numpy(g["cond"])
    bufs = O.schedule_buffers("cosine", 1000)
    x_t = torch.from_numpy(g["step_x_t"])
    for t in (999, 998, 500, 1, 0):
        y = O.p_sample(sd, bufs, x_t, t, cond, torch.from_numpy(g[f"step_{t}_z"]), NH, TD)
        close(y, g[f"step_{t}_out"], rtol=2e-6)
    for T in (1000, 50):
        b = O.schedule_buffers("cosine", T)
        zs = torch.from_numpy(g[f"chain_{T}_z"])         # drawn at t = T-1 .. 1
        y = O.sample(sd, b, cond, torch.from_numpy(g[f"chain_{T}_x_T"]),
                     lambda t: zs[T - 1 - t], NH, TD)
        close(y, g[f"chain_{T}_out"], rtol=5e-5)
        mut, _, _ = O.split_and_binarize(y.numpy(), 8, 24)
        assert np.array_equal(mut, g[f"chain_{T}_mut_mask"])


def test_g5_posterior_table_matches_step_formula(golden_dir):
    """The [T,6] coefficient table reproduces p_sample's update exactly when fed the same eps."""
    g = load(golden_dir, "g5_sampling")
    sd = _small_sd(golden_dir)
    cond = torch.from_numpy(g["cond"])
    bufs = O.schedule_buffers("cosine", 1000)
    coef = O.posterior_coefficients(bufs)
    x = torch.from_numpy(g["step_x_t"])
    for t in (999, 500, 1):
        z = torch.from_numpy(g[f"step_{t}_z"])
        eps = O.unet_forward(sd, x, torch.full((3,), t / 1000), O.condition_embed(sd, cond), NH, TD)
        c = coef[t]
        x0 = (x - c[0] * eps) / c[1]
        y = (c[2] * x0 / c[3] + c[4] * x / c[3]) + c[5] * z
        assert np.array_equal(y.numpy(), O.p_sample(sd, bufs, x, t, cond, z, NH, TD).numpy())


def test_g6_train_epoch(golden_dir):
    g = load(golden_dir, "g6_train_epoch")
    sd = params_only(sd_from(g, "sd0."))
    names = list(sd.keys())
    bufs = O.schedule_buffers("cosine", 1000)
    data, cond, surv = (torch.from_numpy(g[k]) for k in ("data", "cond", "surv"))
    params = [sd[k].clone() for k in names]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    losses = []
    for it in range(4):
        sl = slice(16 * it, 16 * it + 16)
        xm, cm, _ = O.mixup(data[sl], cond[sl], surv[sl], float(g["lam"][it]),
                            torch.from_numpy(g["perm"][it]))
        cur = dict(zip(names, params))
        loss, grads = O.training_loss_and_grads(cur, bufs, xm, cm, torch.from_numpy(g["t"][it]),
                                                torch.from_numpy(g["noise"][it]), NH, TD)
        losses.append(float(loss))
        clipped, _ = O.clip_grad_norm([grads[k] for k in names], 1.0)
        O.adamw_step(params, clipped, m, v, it + 1, lr=1e-4, weight_decay=1e-5)
    close(np.mean(losses), g["avg_loss"], rtol=1e-6)
    for k, p, mm, vv in zip(names, params, m, v):
        close(p, g["sd1." + k], rtol=1e-6, atol=1e-9)
        close(mm, g["exp_avg." + k], rtol=1e-5, atol=1e-12)
        close(vv, g["exp_avg_sq." + k], rtol=1e-5, atol=1e-15)


def test_g7_conditions_and_generate(golden_dir):
    g = load(golden_dir, "g7_generation")
    scen = {
        "early_stage_good_prognosis": dict(survival_time=2000, event_occurred=0, metastasis_at_diagnosis=0),
        "metastatic_poor_prognosis": dict(survival_time=300, event_occurred=1, metastasis_at_diagnosis=1),
        "typical_patient": dict(survival_time=800, event_occurred=0, metastasis_at_diagnosis=0),
    }
    cond_on = ["survival_time", "event_occurred", "metastasis_at_diagnosis"]
    for cd in (3, 4, 2):
        for name, s in scen.items():
            c = O.create_conditions(cond_on, cd, 5, s)
            assert np.array_equal(c, g[f"cd{cd}.{name}"])
    assert O.create_conditions(cond_on, 3, 5, None) is None
    sd = _small_sd(golden_dir)
    T = 20
    bufs = O.schedule_buffers("cosine", T)
    cond = torch.from_numpy(O.create_conditions(cond_on, 3, 6, scen["metastatic_poor_prognosis"]))
    zs = torch.from_numpy(g["gen_z"])
    y = O.sample(sd, bufs, cond, torch.from_numpy(g["gen_x_T"]), lambda t: zs[T - 1 - t], NH, TD)
    mut, expr, path = O.split_and_binarize(y.numpy(), 8, 24)
    assert np.array_equal(mut, g["gen.mutations"]) and mut.dtype == np.float64
    close(expr, g["gen.expression"], rtol=1e-5)
    close(path, g["gen.pathways"], rtol=1e-5)
    assert np.array_equal(cond.numpy(), g["gen.conditions"])


def test_g8_full_shape(golden_dir):
    g = load(golden_dir, "g8_full_shape")
    shapes = O.param_shapes(50, 1900, 50, 3, [256, 512, 256], 128)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 2663952
    assert len(shapes) == 52
    sd = O.init_state_dict(shapes, seed=int(g["init_seed"]))
    x, cond, t, z = (torch.from_numpy(g[k]) for k in ("x", "cond", "t", "z"))
    pred = O.unet_forward(sd, x, t.float() / 1000, O.condition_embed(sd, cond), 3, 128)
    close(pred, g["noise_pred"], rtol=2e-6)
    bufs = O.schedule_buffers("cosine", 1000)
    close(O.p_sample(sd, bufs, x, 640, cond, z, 3, 128), g["p_sample_640"], rtol=2e-6)


def test_g9_validation_metrics(golden_dir):
    """oracle/validation_oracle.py against the reference's BiologicalValidator outputs."""
    from oracle import validation_oracle as V
    g = load(golden_dir, "g9_validation")
    real, synth = g["real"], g["synth"]
    assert abs(V.mmd_rbf(real, synth) - g["mmd"]) < 1e-9
    assert V.mmd_rbf(real, real) < 1e-6 and g["mmd_same"] < 1e-6
    for i in range(100):
        dmax, dmin = V.ks_count_extremes(real[:, i], synth[:, i])
        d, p = V.ks_pvalue(90, 70, dmax, dmin)
        assert abs(d - g["ks_stat"][i]) < 1e-12 and abs(p - g["ks_pvalue"][i]) < 1e-12
    ks = V.ks_summary(real, synth)
    assert abs(ks["ks_test_mean_pvalue"] - g["stat.ks_test_mean_pvalue"]) < 1e-12
    assert ks["ks_test_fraction_significant"] == g["stat.ks_test_fraction_significant"]
    coh = V.pathway_coherence(g["coh_real"], g["coh_synth"], g["coh_member"], 40)
    for k, v in coh.items():
        assert abs(v - g["coh." + k]) < 1e-6, k
    c = [V.pearson(g["me_mut"][:, 0], g["me_pw"][:, 0]), V.pearson(g["me_mut"][:, 1], g["me_pw"][:, 1])]
    assert np.allclose(c, g["me.corr"], atol=1e-12)
    assert V.violation_rate(c, ["negative", "positive"]) == g["me.violation_rate"]


def _cvae_args(g, mode):
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    x, cond, surv, eps = (torch.from_numpy(g[k]) for k in ("x", "cond", "survival", "eps"))
    kw = dict(training=mode == "train", p=0.2)
    if mode == "train":
        m = [torch.from_numpy(g[f"train_mask.{i}"]) for i in range(7)]
        kw.update(enc_masks=m[0:3], dec_masks=m[3:6], surv_mask=m[6])
    return sd, (x, cond, surv, eps), kw


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_g10_cvae(golden_dir, mode):
    """oracle/cvae_oracle.py against the reference's BiologyConstrainedVAE (models/cvae.py)."""
    from oracle import cvae_oracle as V
    g = load(golden_dir, "g10_cvae")
    sd, args, kw = _cvae_args(g, mode)
    new_stats = {}
    out, grads = V.loss_and_grads(sd, *args, new_stats=new_stats, **kw)
    close(out[0].detach(), g[f"{mode}_loss"], rtol=1e-6)
    for k, gr in grads.items():
        # a Linear bias in front of a training-mode BatchNorm has an exactly-zero gradient: fp32 noise ~1e-6 on both sides
        close(gr, g[f"{mode}_grad.{k}"], rtol=2e-5, atol=3e-6)
    if mode == "train":
        for k, v in new_stats.items():
            close(v, g[f"sd_after.{k}"], rtol=1e-6)
    else:
        assert not new_stats
        parts = V.vae_forward(sd, args[0], args[1], args[3], False)
        for name, v in zip(["loss", "x_recon", "mu", "logvar", "recon_loss", "kl_loss"], parts):
            close(v, g[f"eval_parts.{name}"], rtol=1e-6)
        z = torch.from_numpy(g["z_sample"])
        close(V.decode(sd, z, args[1], False), g["eval_sample"], rtol=1e-6)
        close(V.decode(sd, z, args[1], False), g["eval_decode"], rtol=1e-6)
        close(V.encode(sd, args[0], args[1], False)[0], g["eval_encode"], rtol=1e-6)


def test_g9_cooccurrence_and_wasserstein(golden_dir):
    from oracle import validation_oracle as V
    g = load(golden_dir, "g9_validation")
    names = ["TP53", "RB1", "ATRX", "PTEN", "MDM2", "MYC"] + [f"M{i}" for i in range(54)]
    co = V.mutation_cooccurrence(g["co_real"], g["co_synth"], names, ["TP53", "RB1", "ATRX", "DLG2", "PTEN"], [["TP53", "MDM2"]],
                                 g["co_picked"].tolist())
    assert set(co) == {k[3:] for k in g if k.startswith("co.")}
    for k, v in co.items():
        assert abs(v - g["co." + k]) < 1e-10, k
    assert abs(V.wasserstein_pca_mean(g["real"], g["synth"]) - g["stat.wasserstein_distance_mean"]) < 1e-5
