"""GPU: the cVAE mirror (models/cvae.py) through the HIP layer ops, against the reference's own
outputs (tests/golden/g10_cvae.npz) and the CPU oracle (oracle/cvae_oracle.py).

Stated fp32 tolerances: outputs and losses 1e-5 * max|ref|; gradients 5e-5 * max|ref| + 3e-6 (a Linear bias in front of a
training-mode BatchNorm has an exactly-zero gradient, both sides hold fp32 noise there); running statistics 1e-6."""
import numpy as np
import pytest
import torch

from oracle import cvae_oracle as V
from osteosarcoma_diffusionmodel_amd import nn_ops
from osteosarcoma_diffusionmodel_amd.cvae import BiologyConstrainedVAE, ConditionalVAE
from osteosarcoma_diffusionmodel_amd.train import Trainer
from helpers import TAG_DROPOUT, assert_close, load_golden, philox_block

pytestmark = pytest.mark.gpu
CONF = {"model": {"latent_dim": 16, "hidden_dims": [32, 64, 32], "gnn": {"dropout": 0.2},
                  "constraints": {"pathway_coherence_weight": 1.0, "mutation_expression_weight": 0.5, "survival_prediction_weight": 0.3}}}


def dev(a):
    return torch.from_numpy(np.asarray(a)).cuda()


def golden_model(g):
    m = BiologyConstrainedVAE(8, 24, 8, 3, CONF)
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    assert list(m.state_dict().keys()) == list(sd.keys())          # the reference's key names, in its order
    m.load_state_dict(sd)
    return m.cuda()


def test_eval_mode_vs_reference(golden_dir):
    g = load_golden(golden_dir, "g10_cvae")
    m = golden_model(g).eval()
    x, cond, surv, eps, zs = (dev(g[k]) for k in ("x", "cond", "survival", "eps", "z_sample"))
    loss = m(x, cond, surv, eps=eps)
    loss.backward()
    assert_close(loss.item(), g["eval_loss"], 1e-5, what="loss")
    for k, p in m.named_parameters():
        assert_close(p.grad.cpu(), g[f"eval_grad.{k}"], 5e-5, atol=3e-6, what=f"grad {k}")
    with torch.no_grad():
        parts = m.vae(x, cond, return_parts=True, eps=eps)
    for name, v in zip(["loss", "x_recon", "mu", "logvar", "recon_loss", "kl_loss"], parts):
        assert_close(v.cpu(), g[f"eval_parts.{name}"], 1e-5, what=name)
    assert_close(m.sample(cond, num_samples=6, z=zs).cpu(), g["eval_sample"], 1e-5, what="sample")
    assert_close(m.vae.encode(x, cond).cpu(), g["eval_encode"], 1e-5, what="encode")
    assert_close(m.vae.decode(zs, cond).cpu(), g["eval_decode"], 1e-5, what="decode")
    assert m.sample(cond, num_samples=6).shape == (6, 40)           # torch.randn latent, as the reference
    # eval mode leaves the running statistics alone
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            assert np.array_equal(v.cpu().numpy(), g[f"sd.{k}"]), k


def test_train_mode_vs_reference(golden_dir):
    g = load_golden(golden_dir, "g10_cvae")
    m = golden_model(g).train()
    x, cond, surv, eps = (dev(g[k]) for k in ("x", "cond", "survival", "eps"))
    masks = [dev(g[f"train_mask.{i}"]) for i in range(7)]
    loss = m(x, cond, surv, eps=eps, dropout_masks=(masks[0:3], masks[3:6]), survival_mask=masks[6])
    loss.backward()
    assert_close(loss.item(), g["train_loss"], 1e-5, what="loss")
    for k, p in m.named_parameters():
        assert_close(p.grad.cpu(), g[f"train_grad.{k}"], 5e-5, atol=3e-6, what=f"grad {k}")
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            assert_close(v.cpu(), g[f"sd_after.{k}"], 1e-6, what=k)


def test_full_shape_vs_oracle():
    """D = 2000, hidden [256, 512, 256], latent 128, batch 192: train-mode loss and gradients against the oracle,
    dropout masks drawn by the library's Philox stream (checked against the numpy restatement)."""
    conf = {"model": {"latent_dim": 128, "hidden_dims": [256, 512, 256], "gnn": {"dropout": 0.2}, "constraints": CONF["model"]["constraints"]}}
    torch.manual_seed(1)
    m = BiologyConstrainedVAE(50, 1900, 50, 3, conf).cuda().train()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(2)
    B = 192
    x = torch.randn(B, 2000, generator=g)
    cond = torch.randn(B, 3, generator=g)
    surv = torch.randn(B, generator=g)
    eps = torch.randn(B, 128, generator=g)
    seed = 987654321
    widths = [256, 512, 256, 256, 512, 256, 128]
    tags = [0x10, 0x11, 0x12, 0x20, 0x21, 0x22, 0x30]
    masks = []
    for w, tg in zip(widths, tags):
        u = (philox_block(seed, B, w, 0, TAG_DROPOUT + tg) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
        masks.append(torch.from_numpy((u >= np.float32(0.2)).astype(np.float32)))
    new_stats = {}
    out, grads = V.loss_and_grads(sd, x, cond, surv, eps, training=True, enc_masks=masks[0:3], dec_masks=masks[3:6], surv_mask=masks[6],
                                  p=0.2, new_stats=new_stats)
    loss = m(x.cuda(), cond.cuda(), surv.cuda(), eps=eps.cuda(), seed=seed)
    loss.backward()
    assert_close(loss.item(), out[0].item(), 1e-5, what="loss")
    for k, p in m.named_parameters():
        assert_close(p.grad.cpu(), grads[k], 5e-5, atol=3e-6, what=f"grad {k}")
    for k, v in new_stats.items():
        assert_close(m.state_dict()[k].cpu(), v, 1e-5, what=k)


def test_conditional_vae_alone_and_errors():
    torch.manual_seed(0)
    m = ConditionalVAE(8, 24, 8, 3, CONF).cuda()
    x, c = torch.randn(5, 40).cuda(), torch.randn(5, 3).cuda()
    loss = m(x, c)
    assert loss.dim() == 0 and torch.isfinite(loss)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    with pytest.raises(ValueError):                                 # torch: "Expected more than 1 value per channel when training"
        m(x[:1], c[:1])
    m.eval()
    assert torch.isfinite(m(x[:1], c[:1]))                          # running statistics: a single row is fine
    with pytest.raises(RuntimeError):
        m(x.cpu(), c.cpu())                                         # no CPU fallback
    with pytest.raises(RuntimeError):
        nn_ops.linear(x, None, torch.randn(4, 39).cuda(), torch.zeros(4).cuda())   # shape mismatch


def test_trainer_runs_cvae_epoch(tmp_path):
    """BASELINE config 1 (plumbing): synthetic n = 100, batch 16, one epoch with the reference's Trainer flow."""
    rng = np.random.default_rng(42)
    n, D = 100, 40
    data = rng.standard_normal((n, D)).astype(np.float32)
    data[:, :8] = rng.integers(0, 2, (n, 8))
    cond = rng.standard_normal((n, 3)).astype(np.float32)
    surv = rng.standard_normal(n).astype(np.float32)
    rows = [{"data": torch.from_numpy(data[i]), "conditions": torch.from_numpy(cond[i]), "survival": torch.tensor(surv[i])} for i in range(n)]
    tl = torch.utils.data.DataLoader(rows[:80], batch_size=16, shuffle=True, drop_last=True)
    vl = torch.utils.data.DataLoader(rows[80:], batch_size=16)
    conf = dict(CONF)
    conf["training"] = {"learning_rate": 1e-3, "weight_decay": 1e-5, "patience": 100, "min_delta": 1e-4, "augmentation": {"mixup_alpha": 0.2},
                        "save_dir": str(tmp_path), "num_epochs": 2, "save_frequency": 1, "val_split": 0.2, "random_seed": 42, "batch_size": 16}
    torch.manual_seed(0)
    np.random.seed(0)
    m = BiologyConstrainedVAE(8, 24, 8, 3, conf)
    tr = Trainer(m, tl, vl, conf, device="cuda")
    hist = tr.train()
    assert len(hist["train_loss"]) == 2 and all(np.isfinite(hist["train_loss"])) and all(np.isfinite(hist["val_loss"]))
    assert hist["train_loss"][1] < hist["train_loss"][0]
    ck = torch.load(tmp_path / "best_model.pt", map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "val_loss", "config"}
    assert "vae.encoder.mlp.1.running_mean" in ck["model_state_dict"]
    assert tr.global_step == 10


def test_config1_cvae_epoch_at_its_stated_shape(tmp_path):
    """BASELINE config 1 at the shape SURVEY section 8(d) states: the cVAE (models/cvae.py:222-346) behind the reference's
    Trainer dispatch (utils/train.py:233-234) on a synthetic "TARGET-OS" cohort generated as QUICKSTART.md:217-241 does but with
    dims 50 / 1900 / 50 -- mutations randint(0, 2), expression and pathways N(0, 1) (pathways z-scored as prepare_data does),
    clinical survival_days ~ randint(100, 2000), event_occurred, age_years ~ U(10, 18), metastasis_at_diagnosis, numpy seed 42;
    the four clinical columns give condition_dim 4 (utils/train.py:395-398); n = 100, val_split 0.2, batch 16 -> 80 training
    rows, 5 steps per epoch (drop_last).  One epoch must run with a finite loss, a second one must lower it; checkpoint keys
    as utils/train.py:278-284.  Runs through the OsteosarcomaDataset / DataLoader objects of the pipeline (and hence the
    device-resident epoch path)."""
    from osteosarcoma_diffusionmodel_amd.train import OsteosarcomaDataset
    np.random.seed(42)
    n = 100
    ids = [f"TARGET-40-{i:04d}" for i in range(n)]
    import pandas as pd
    mut = pd.DataFrame(np.random.randint(0, 2, (n, 50)), index=ids, columns=[f"GENE_{i}" for i in range(50)])
    expr = pd.DataFrame(np.random.randn(n, 1900), index=ids, columns=[f"EXPR_{i}" for i in range(1900)])
    path = pd.DataFrame(np.random.randn(n, 50), index=ids, columns=[f"PATHWAY_{i}" for i in range(50)])
    path = (path - path.mean()) / (path.std() + 1e-8)
    clinical = pd.DataFrame({"submitter_id": ids, "survival_days": np.random.randint(100, 2000, n), "event_occurred": np.random.randint(0, 2, n),
                             "age_years": np.random.uniform(10, 18, n), "metastasis_at_diagnosis": np.random.randint(0, 2, n)})
    clinical["survival_days_norm"] = (clinical["survival_days"] - clinical["survival_days"].mean()) / (clinical["survival_days"].std() + 1e-8)
    feats = ["survival_days_norm", "event_occurred", "age_years", "metastasis_at_diagnosis"]
    ds = OsteosarcomaDataset(mut, expr, path, clinical, feats)
    assert ds.data.shape == (100, 2000) and ds.conditions.shape == (100, 4)
    tr_ds, va_ds = torch.utils.data.random_split(ds, [80, 20], generator=torch.Generator().manual_seed(42))
    tl = torch.utils.data.DataLoader(tr_ds, batch_size=16, shuffle=True, num_workers=0, drop_last=True)
    vl = torch.utils.data.DataLoader(va_ds, batch_size=16, shuffle=False, num_workers=0)
    # the reference's own model section (config/config.yaml:34-61): latent 128, hidden [256, 512, 256], dropout 0.2
    conf = {"model": {"latent_dim": 128, "hidden_dims": [256, 512, 256], "gnn": {"dropout": 0.2}, "constraints": CONF["model"]["constraints"]}}
    conf["training"] = {"learning_rate": 1e-3, "weight_decay": 1e-5, "patience": 100, "min_delta": 1e-4, "augmentation": {"mixup_alpha": 0.2},
                        "save_dir": str(tmp_path), "num_epochs": 2, "save_frequency": 1, "val_split": 0.2, "random_seed": 42, "batch_size": 16}
    torch.manual_seed(0)
    m = BiologyConstrainedVAE(50, 1900, 50, 4, conf)
    tr = Trainer(m, tl, vl, conf, device="cuda")
    assert len(tl) == 5
    hist = tr.train()
    assert tr.resident is True                                     # the loaders are of prepare_data's shape: replayed from HBM
    assert tr.global_step == 10
    assert len(hist["train_loss"]) == 2 and all(np.isfinite(hist["train_loss"])) and all(np.isfinite(hist["val_loss"]))
    assert hist["train_loss"][1] < hist["train_loss"][0]
    ck = torch.load(tmp_path / "best_model.pt", map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "val_loss", "config"}
    assert ck["model_state_dict"]["vae.encoder.mlp.0.weight"].shape[1] == 2000 + 4
    # the same loaders through the reference's per-batch host hand-over: same number of steps, finite
    conf2 = dict(conf)
    conf2["training"] = dict(conf["training"], resident_dataset=False, save_dir=str(tmp_path / "b"), num_epochs=1)
    torch.manual_seed(0)
    m2 = BiologyConstrainedVAE(50, 1900, 50, 4, conf2)
    tr2 = Trainer(m2, tl, vl, conf2, device="cuda")
    h2 = tr2.train()
    assert tr2.resident is False and tr2.global_step == 5 and np.isfinite(h2["train_loss"][0])
