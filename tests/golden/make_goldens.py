#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE on CPU.

Run once, in the build container only (where /root/reference is mounted):

    python tests/golden/make_goldens.py

It imports ``models.diffusion`` / ``utils.train`` / ``utils.generate`` from
/root/reference (read-only, never copied), feeds them seeded inputs with every
random draw injected (``torch.randint`` / ``randn_like`` / ``randn`` /
``F.dropout`` / ``np.random.beta`` / ``torch.randperm`` are wrapped so the draws
are recorded), and writes inputs + the reference's outputs as small ``.npz``
files.  The fixtures are DATA only; no reference source is stored.

``torch_geometric`` is absent offline and is only needed by a class the reference
never instantiates (models/diffusion.py:14-88), so an empty stand-in module is
placed in ``sys.modules`` for the import.
"""
import os
import sys
import tempfile
import types
from collections import deque
from pathlib import Path

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import torch
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")

# ---- import the reference -------------------------------------------------
_pyg = types.ModuleType("torch_geometric")
_pyg_nn = types.ModuleType("torch_geometric.nn")
_pyg_nn.GATConv = object
_pyg_nn.global_mean_pool = lambda *a, **k: None
_pyg.nn = _pyg_nn
sys.modules.setdefault("torch_geometric", _pyg)
sys.modules.setdefault("torch_geometric.nn", _pyg_nn)
sys.path.insert(0, str(REF))

from models.diffusion import BiologyAwareDiffusionModel, TimeEmbedding  # noqa: E402
import utils.train as ref_train  # noqa: E402
import utils.generate as ref_generate  # noqa: E402

torch.set_num_threads(4)


def cfg(hidden, T=1000, schedule="cosine", p=0.2, cond_on=None):
    return {
        "model": {
            "latent_dim": 128, "hidden_dims": list(hidden), "gnn": {"dropout": p},
            "diffusion": {"num_steps": T, "beta_schedule": schedule},
            "condition_on": cond_on or ["survival_time", "event_occurred", "metastasis_at_diagnosis"],
        },
        "training": {
            "learning_rate": 1e-4, "weight_decay": 1e-5, "patience": 100, "min_delta": 1e-4,
            "augmentation": {"mixup_alpha": 0.2}, "save_dir": tempfile.mkdtemp(prefix="osd_gold_"),
            "num_epochs": 1, "save_frequency": 10, "val_split": 0.2, "random_seed": 42, "batch_size": 16,
        },
    }


def npd(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


class Inject:
    """Context manager that replaces torch's RNG entry points with recorded queues."""

    def __init__(self, randint=None, randn_like=None, randn=None, drop_seed=None):
        self.q_randint = deque(randint or [])
        self.q_randn_like = deque(randn_like or [])
        self.q_randn = deque(randn or [])
        self.drop_gen = None if drop_seed is None else torch.Generator().manual_seed(drop_seed)
        self.masks = []

    def __enter__(self):
        self._o = (torch.randint, torch.randn_like, torch.randn, F.dropout)
        inj = self

        def randint(*a, **k):
            return inj.q_randint.popleft().clone()

        def randn_like(x, *a, **k):
            t = inj.q_randn_like.popleft()
            assert t.shape == x.shape
            return t.clone()

        def randn(*a, **k):
            return inj.q_randn.popleft().clone()

        def dropout(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            keep = (torch.rand(x.shape, generator=inj.drop_gen) >= p).to(x.dtype)
            inj.masks.append(keep)
            return x * (keep / (1.0 - p))

        torch.randint, torch.randn_like, torch.randn = randint, randn_like, randn
        F.dropout = dropout
        torch.nn.functional.dropout = dropout
        return self

    def __exit__(self, *exc):
        torch.randint, torch.randn_like, torch.randn, F.dropout = self._o
        torch.nn.functional.dropout = self._o[3]
        return False


def save(name, **arrays):
    path = HERE / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"wrote {path.name}: {path.stat().st_size / 1024:.1f} KiB")


# ---- G1 schedule buffers ---------------------------------------------------
def g1():
    out = {}
    for sched in ("cosine", "linear"):
        for T in (1000, 50):
            m = BiologyAwareDiffusionModel(8, 24, 8, 3, cfg([32, 64, 32], T=T, schedule=sched))
            for k in ("betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"):
                out[f"{sched}_{T}_{k}"] = getattr(m, k).numpy()
    save("g1_schedule", **out)


# ---- G2 time embedding -----------------------------------------------------
def g2():
    te = TimeEmbedding(128)
    tt = torch.tensor([0, 1, 2, 10, 100, 250, 499, 500, 501, 750, 900, 990, 997, 998, 999, 333],
                      dtype=torch.int64)
    t_train = tt.float() / 1000                                 # models/diffusion.py:367
    t_samp = torch.stack([torch.full((1,), int(t) / 1000)[0] for t in tt])   # :392
    save("g2_time_embedding", t_index=tt.numpy(), t_norm_train=t_train.numpy(),
         t_norm_sample=t_samp.numpy(), emb=te(t_train).numpy())


# ---- small model shared by G3..G6 -------------------------------------------
SM = dict(mutation_dim=8, expression_dim=24, pathway_dim=8, condition_dim=3)
SM_H = [32, 64, 32]


def small_model(T=1000, p=0.2, seed=0):
    torch.manual_seed(seed)
    m = BiologyAwareDiffusionModel(config=cfg(SM_H, T=T, p=p), **SM)
    # make GroupNorm affine non-trivial so the fixtures exercise gamma/beta
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        for k, v in m.named_parameters():
            parts = k.split(".")
            if parts[-2] in ("1", "5") and v.dim() == 1 and parts[1] != "input_proj":
                if parts[-1] == "weight":
                    v.copy_(1.0 + 0.3 * torch.randn(v.shape, generator=g))
                else:
                    v.copy_(0.2 * torch.randn(v.shape, generator=g))
    return m


def g3_g4():
    g = torch.Generator().manual_seed(7)
    B, D = 4, 40
    x = torch.randn(B, D, generator=g)
    cond = torch.randn(B, 3, generator=g)
    t = torch.tensor([0, 17, 500, 999])
    noise = torch.randn(B, D, generator=g)

    m = small_model()
    out = {f"sd.{k}": v for k, v in npd(m.state_dict()).items()}
    out.update(x=x.numpy(), cond=cond.numpy(), t=t.numpy(), noise=noise.numpy())

    # eval-mode denoiser forward with every block output captured
    m.eval()
    taps = {}
    hooks = []
    for name, mod in m.unet.named_modules():
        if name in ("encoder.0", "encoder.1", "bottleneck", "decoder.0", "decoder.1",
                    "input_proj", "time_proj", "cond_proj"):
            hooks.append(mod.register_forward_hook(
                lambda mod, i, o, name=name: taps.__setitem__(name, o.detach().clone())))
    with torch.no_grad():
        c_emb = m.condition_embed(cond)
        pred = m.unet(x, t.float() / 1000, c_emb)
    for h in hooks:
        h.remove()
    out["eval_c_emb"] = c_emb.numpy()
    out["eval_noise_pred"] = pred.numpy()
    for k, v in taps.items():
        out[f"eval_tap.{k}"] = v.numpy()

    # q_sample with given noise
    with torch.no_grad():
        x_t, _ = m.q_sample(x, t, noise)
    out["q_sample_x_t"] = x_t.numpy()

    # training forward, eval mode (no dropout): loss + noise_pred + all grads
    m.zero_grad()
    with Inject(randint=[t], randn_like=[noise]):
        loss = m(x, cond, return_loss=True)
    loss.backward()
    out["eval_loss"] = loss.detach().numpy()
    for k, v in m.named_parameters():
        out[f"eval_grad.{k}"] = v.grad.numpy().copy()
    with Inject(randint=[t], randn_like=[noise]):
        with torch.no_grad():
            out["eval_forward_noise_pred"] = m(x, cond, return_loss=False).numpy()

    # training forward, train mode with recorded dropout masks
    m.train()
    m.zero_grad()
    with Inject(randint=[t], randn_like=[noise], drop_seed=11) as inj:
        loss = m(x, cond, return_loss=True)
    loss.backward()
    out["train_loss"] = loss.detach().numpy()
    for i, mk in enumerate(inj.masks):
        out[f"train_mask.{i}"] = mk.numpy()
    for k, v in m.named_parameters():
        out[f"train_grad.{k}"] = v.grad.numpy().copy()
    save("g3g4_small_model", **out)


def g5():
    g = torch.Generator().manual_seed(21)
    N, D = 3, 40
    cond = torch.randn(N, 3, generator=g)
    m = small_model()
    m.eval()
    out = {"cond": cond.numpy()}
    # single steps
    x_t = torch.randn(N, D, generator=g)
    out["step_x_t"] = x_t.numpy()
    for t in (999, 998, 500, 1, 0):
        z = torch.randn(N, D, generator=g)
        with Inject(randn_like=[z]):
            y = m.p_sample(x_t, t, cond)
        out[f"step_{t}_z"] = z.numpy()
        out[f"step_{t}_out"] = y.numpy()
    # full chains
    for T in (1000, 50):
        mt = small_model(T=T)
        mt.eval()
        x_T = torch.randn(N, D, generator=g)
        zs = [torch.randn(N, D, generator=g) for _ in range(T - 1)]   # drawn at t = T-1 .. 1
        with Inject(randn=[x_T], randn_like=list(zs)):
            y = mt.sample(cond, num_samples=N)
        out[f"chain_{T}_x_T"] = x_T.numpy()
        out[f"chain_{T}_z"] = torch.stack(zs).numpy()
        out[f"chain_{T}_out"] = y.numpy()
        out[f"chain_{T}_mut_mask"] = (y.numpy()[:, :8] > 0.5).astype(float)
    save("g5_sampling", **out)


def g6():
    """One Trainer.train_epoch on 64 rows, batch 16, mixup on, dropout off (p=0)."""
    n, D = 64, 40
    g = torch.Generator().manual_seed(5)
    data = torch.randn(n, D, generator=g)
    data[:, :8] = (data[:, :8] > 0).float()
    cond = torch.randn(n, 3, generator=g)
    surv = torch.rand(n, generator=g) * 1000

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return n

        def __getitem__(self, i):
            return {"data": data[i], "conditions": cond[i], "survival": surv[i]}

    c = cfg(SM_H, p=0.0)
    torch.manual_seed(0)
    m = small_model(p=0.0)
    sd0 = npd(m.state_dict())
    loader = torch.utils.data.DataLoader(DS(), batch_size=16, shuffle=False, drop_last=True)
    tr = ref_train.Trainer(m, loader, loader, c, device="cpu")

    lams, perms, ts, noises = [], [], [], []
    gg = torch.Generator().manual_seed(9)
    for _ in range(4):
        ts.append(torch.randint(0, 1000, (16,), generator=gg))
        noises.append(torch.randn(16, D, generator=gg))
    o_beta, o_perm = np.random.beta, torch.randperm
    rs = np.random.RandomState(3)

    def beta(a, b):
        v = float(rs.beta(a, b))
        lams.append(v)
        return v

    def randperm(k, *a, **kw):
        p = o_perm(k, generator=gg)
        perms.append(p)
        return p

    np.random.beta, torch.randperm = beta, randperm
    try:
        with Inject(randint=list(ts), randn_like=list(noises)):
            avg = tr.train_epoch()
    finally:
        np.random.beta, torch.randperm = o_beta, o_perm
    out = {f"sd0.{k}": v for k, v in sd0.items()}
    out.update({f"sd1.{k}": v for k, v in npd(m.state_dict()).items()})
    st = tr.optimizer.state_dict()["state"]
    names = [k for k, _ in m.named_parameters()]
    for i, k in enumerate(names):
        out[f"exp_avg.{k}"] = st[i]["exp_avg"].numpy()
        out[f"exp_avg_sq.{k}"] = st[i]["exp_avg_sq"].numpy()
    out.update(data=data.numpy(), cond=cond.numpy(), surv=surv.numpy(),
               lam=np.asarray(lams), perm=torch.stack(perms).numpy(),
               t=torch.stack(ts).numpy(), noise=torch.stack(noises).numpy(),
               avg_loss=np.asarray(avg))
    save("g6_train_epoch", **out)


def g7():
    """create_conditions for the three config.yaml scenarios at cond_dim 3 and 4."""
    import yaml
    conf = yaml.safe_load(open(REF / "config" / "config.yaml"))
    out = {}
    for cd in (3, 4, 2):
        m = BiologyAwareDiffusionModel(8, 24, 8, cd, cfg(SM_H))
        gen = ref_generate.SyntheticPatientGenerator(m, conf, device="cpu")
        for sc in conf["generation"]["scenarios"]:
            out[f"cd{cd}.{sc['name']}"] = gen.create_conditions(5, sc["conditions"]).numpy()
    # generate(): split + binarise on a T=20 model with injected noise
    T = 20
    m = small_model(T=T)
    gen = ref_generate.SyntheticPatientGenerator(m, conf, device="cpu")
    g = torch.Generator().manual_seed(77)
    N = 6
    x_T = torch.randn(N, 40, generator=g)
    zs = [torch.randn(N, 40, generator=g) for _ in range(T - 1)]
    with Inject(randn=[x_T], randn_like=list(zs)):
        res = gen.generate(N, conf["generation"]["scenarios"][1]["conditions"])
    out["gen_x_T"] = x_T.numpy()
    out["gen_z"] = torch.stack(zs).numpy()
    for k, v in res.items():
        out[f"gen.{k}"] = np.asarray(v)
    save("g7_generation", **out)


def g8():
    """Full shape (D=2000, hidden [256,512,256]) single eval forward + one p_sample on 4
    rows.  Weights are NOT stored (10.7 MB): the model is loaded with the oracle's own
    seeded initialiser, so the fixture pins 'reference(weights W) == oracle(weights W)'."""
    sys.path.insert(0, str(HERE.parent.parent))
    from oracle import diffusion_oracle as O
    shapes = O.param_shapes(50, 1900, 50, 3, [256, 512, 256], 128)
    sd = O.init_state_dict(shapes, seed=1234)
    m = BiologyAwareDiffusionModel(50, 1900, 50, 3, cfg([256, 512, 256]))
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all("alphas" in k or "betas" in k for k in missing.missing_keys)
    m.eval()
    g = torch.Generator().manual_seed(99)
    x = torch.randn(4, 2000, generator=g)
    cond = torch.randn(4, 3, generator=g)
    t = torch.tensor([3, 400, 777, 999])
    z = torch.randn(4, 2000, generator=g)
    with torch.no_grad():
        pred = m.unet(x, t.float() / 1000, m.condition_embed(cond))
        with Inject(randn_like=[z]):
            step = m.p_sample(x, 640, cond)
    save("g8_full_shape", x=x.numpy(), cond=cond.numpy(), t=t.numpy(), z=z.numpy(),
         init_seed=np.asarray(1234), noise_pred=pred.numpy(), p_sample_640=step.numpy())


def g9():
    """Validation metrics (utils/validation.py): MMD, KS, pathway coherence, mutation-expression sign check."""
    import pandas as pd
    import yaml
    from utils.validation import BiologicalValidator
    conf = yaml.safe_load(open(REF / "config" / "config.yaml"))
    val = BiologicalValidator(conf)
    rs = np.random.RandomState(11)
    out = {}
    # MMD + KS on [n, D] arrays (real vs shifted synthetic)
    real = rs.randn(90, 130).astype(np.float32)
    synth = (rs.randn(70, 130) * 1.1 + 0.15).astype(np.float32)
    synth[:, :20] = real[rs.randint(0, 90, 70), :20]            # ties between the two samples
    out["real"], out["synth"] = real, synth
    out["mmd"] = np.float64(val.compute_mmd(real, synth))
    out["mmd_same"] = np.float64(val.compute_mmd(real, real))
    st = val.statistical_tests(real, synth)
    for k in ("ks_test_mean_pvalue", "ks_test_fraction_significant", "mmd"):
        out[f"stat.{k}"] = np.float64(st[k])
    from scipy import stats as sps
    ks = [sps.ks_2samp(real[:, i], synth[:, i]) for i in range(100)]
    out["ks_stat"] = np.array([k.statistic for k in ks])
    out["ks_pvalue"] = np.array([k.pvalue for k in ks])
    # pathway coherence: 12 pathways over 40 genes (first 10 used; one with < 3 genes present is skipped)
    genes = [f"G{i}" for i in range(40)]
    base = rs.randn(90, 6)
    load = rs.randn(6, 40) * (rs.rand(6, 40) < 0.3)
    real_e = (base @ load + 0.7 * rs.randn(90, 40)).astype(np.float32)
    synth_e = (rs.randn(70, 6) @ load + 0.9 * rs.randn(70, 40)).astype(np.float32)
    member = (rs.rand(45, 12) < 0.25).astype(int)                # 45 genes listed, 5 of them absent from the data
    member[:, 3] = 0
    member[[1, 2], 3] = 1                                        # pathway 3: only 2 genes -> skipped
    pgm = pd.DataFrame(member, index=[f"G{i}" for i in range(45)], columns=[f"P{i}" for i in range(12)])
    coh = val.validate_pathway_coherence(pd.DataFrame(real_e, columns=genes), pd.DataFrame(synth_e, columns=genes), pgm)
    out["coh_real"], out["coh_synth"], out["coh_member"] = real_e, synth_e, member
    for k, v in coh.items():
        out[f"coh.{k}"] = np.float64(v)
    # mutation-expression sign rules of config.yaml:110-116
    mut = pd.DataFrame((rs.rand(70, 3) < 0.4).astype(float), columns=["TP53", "MYC", "RB1"])
    pw = pd.DataFrame(rs.randn(70, 2), columns=["HALLMARK_P53_PATHWAY", "HALLMARK_MYC_TARGETS_V1"])
    pw["HALLMARK_P53_PATHWAY"] += 0.8 * mut["TP53"]              # positive although "negative" is required -> violation
    pw["HALLMARK_MYC_TARGETS_V1"] += 0.8 * mut["MYC"]            # positive as required
    me = val.validate_mutation_expression_correlation(mut, None, pw)
    out["me_mut"], out["me_pw"] = mut.values, pw.values
    out["me.violation_rate"] = np.float64(me["mutation_expression_violation_rate"])
    out["me.corr"] = np.array([mut["TP53"].corr(pw["HALLMARK_P53_PATHWAY"]), mut["MYC"].corr(pw["HALLMARK_MYC_TARGETS_V1"])])
    out["stat.wasserstein_distance_mean"] = np.float64(st["wasserstein_distance_mean"])
    # mutation co-occurrence (utils/validation.py:27-121): 60 genes incl. 4 of the 5 driver genes and the exclusive pair
    names = ["TP53", "RB1", "ATRX", "PTEN", "MDM2", "MYC"] + [f"M{i}" for i in range(54)]
    pr = rs.rand(60) * 0.5 + 0.1
    real_m = (rs.rand(90, 60) < pr).astype(float)
    synth_m = (rs.rand(70, 60) < np.clip(pr + 0.1 * rs.randn(60), 0.05, 0.9)).astype(float)
    synth_m[:, 4] = np.where(synth_m[:, 0] == 1, (rs.rand(70) < 0.1).astype(float), synth_m[:, 4])   # TP53 / MDM2 mostly exclusive
    rm, sm = pd.DataFrame(real_m, columns=names), pd.DataFrame(synth_m, columns=names)
    np.random.seed(123)
    co = val.validate_mutation_cooccurrence(rm, sm)
    np.random.seed(123)
    picked = np.random.choice(pd.Index(names), size=50, replace=False)
    out["co_real"], out["co_synth"] = real_m, synth_m
    out["co_picked"] = np.array([names.index(g) for g in picked])
    for k, v in co.items():
        out[f"co.{k}"] = np.float64(v)
    # validate_all on the pieces above (70 synthetic / 90 real rows everywhere)
    np.random.seed(123)
    real_pw = rs.randn(90, 2)
    out["all_real_pw"] = real_pw
    allr = val.validate_all(rm, pd.DataFrame(real_e, columns=genes), pd.DataFrame(real_pw, columns=pw.columns),
                            sm, pd.DataFrame(synth_e, columns=genes), pw, pgm)
    for k, v in allr.items():
        out[f"all.{k}"] = np.float64(v)
    save("g9_validation", **out)



def g10():
    """cVAE (models/cvae.py): BiologyConstrainedVAE forward/backward in train and eval mode, sample/encode/decode."""
    from models.cvae import BiologyConstrainedVAE
    conf = {"model": {"latent_dim": 16, "hidden_dims": [32, 64, 32], "gnn": {"dropout": 0.2},
                      "constraints": {"pathway_coherence_weight": 1.0, "mutation_expression_weight": 0.5,
                                      "survival_prediction_weight": 0.3}}}
    torch.manual_seed(3)
    m = BiologyConstrainedVAE(8, 24, 8, 3, conf)
    g = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.copy_(1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
                mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g))
    B, D, Lz = 6, 40, 16
    x = torch.randn(B, D, generator=g)
    x[:, :8] = (x[:, :8] > 0).float()
    cond = torch.randn(B, 3, generator=g)
    surv = torch.randn(B, generator=g)
    eps = torch.randn(B, Lz, generator=g)
    zs = torch.randn(B, Lz, generator=g)
    out = {"x": x.numpy(), "cond": cond.numpy(), "survival": surv.numpy(), "eps": eps.numpy(), "z_sample": zs.numpy()}
    out.update({f"sd.{k}": v for k, v in npd(m.state_dict()).items()})

    # eval mode (running statistics, no dropout)
    m.eval()
    with Inject(randn_like=[eps, eps], randn=[zs]):
        loss = m(x, cond, surv)
        m.zero_grad()
        loss.backward()
        out["eval_loss"] = loss.detach().numpy()
        for k, p_ in m.named_parameters():
            out[f"eval_grad.{k}"] = p_.grad.detach().numpy().copy()
        with torch.no_grad():
            parts = m.vae(x, cond, return_parts=True)
        for name, v in zip(["loss", "x_recon", "mu", "logvar", "recon_loss", "kl_loss"], parts):
            out[f"eval_parts.{name}"] = v.detach().numpy()
        out["eval_sample"] = m.sample(cond, num_samples=B).numpy()
        out["eval_encode"] = m.vae.encode(x, cond).numpy()
        out["eval_decode"] = m.vae.decode(zs, cond).numpy()

    # train mode (batch statistics, running-stat update, dropout)
    m.train()
    with Inject(randn_like=[eps], drop_seed=29) as inj:
        loss = m(x, cond, surv)
        m.zero_grad()
        loss.backward()
        out["train_loss"] = loss.detach().numpy()
        for k, p_ in m.named_parameters():
            out[f"train_grad.{k}"] = p_.grad.detach().numpy().copy()
        for i, mk in enumerate(inj.masks):
            out[f"train_mask.{i}"] = mk.numpy()
        assert len(inj.masks) == 7, len(inj.masks)
    out.update({f"sd_after.{k}": v for k, v in npd(m.state_dict()).items() if "running" in k or "num_batches" in k})
    save("g10_cvae", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        for name in sys.argv[1:]:
            globals()[name]()
    else:
        g1(); g2(); g3_g4(); g5(); g6(); g7(); g8(); g9(); g10()
    # leave nothing behind in the read-only reference tree
    for pc in REF.rglob("__pycache__"):
        print("WARNING: bytecode dir appeared:", pc)
