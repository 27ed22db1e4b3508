"""GPU: training path (fused forward+backward, mixup, clip+AdamW, Trainer) against the golden
fixtures of the reference and the CPU oracle.

Stated fp32 tolerances: loss 1e-5 relative; each gradient tensor max|d| <= 5e-5 * max|ref|
(+1e-8 abs; reductions over the batch run in a different order and through float atomics);
parameters after an epoch of AdamW steps 2e-5 * max|ref|."""
import copy

import numpy as np
import pytest
import torch

from oracle import diffusion_oracle as O
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
from osteosarcoma_diffusionmodel_amd.train import FlatParams, FusedAdamW, MixupAugmentation, Trainer
from helpers import FULL, FULL_H, SM, SM_H, assert_close, config, golden_small_sd, load_golden, small_model

pytestmark = pytest.mark.gpu
GRAD_RTOL = 5e-5


def dev(a):
    return torch.from_numpy(np.asarray(a)).cuda()


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_loss_and_grads_vs_reference(golden_dir, mode):
    g = load_golden(golden_dir, "g3g4_small_model")
    m = small_model(golden_dir)
    masks = None
    if mode == "train":
        m.train()
        masks = [dev(g[f"train_mask.{i}"]) for i in range(5)]
    loss = m(dev(g["x"]), dev(g["cond"]), return_loss=True, t=dev(g["t"]), noise=dev(g["noise"]), dropout_masks=masks)
    assert loss.dim() == 0
    loss.backward()
    assert_close(loss.item(), g[f"{mode}_loss"], 1e-5, what="loss")
    for k, p in m.named_parameters():
        assert_close(p.grad.cpu(), g[f"{mode}_grad.{k}"], GRAD_RTOL, atol=1e-8, what=f"grad {k}")
    # return_loss=False gives the predicted noise for the same draws
    pred = m(dev(g["x"]), dev(g["cond"]), return_loss=False, t=dev(g["t"]), noise=dev(g["noise"]), dropout_masks=masks)
    if mode == "eval":
        assert_close(pred.cpu(), g["eval_forward_noise_pred"], 1e-5, what="noise_pred")


def test_backward_scales_with_upstream_gradient(golden_dir):
    g = load_golden(golden_dir, "g3g4_small_model")
    m = small_model(golden_dir)
    args = dict(t=dev(g["t"]), noise=dev(g["noise"]))
    (3.0 * m(dev(g["x"]), dev(g["cond"]), **args)).backward()
    assert_close(m.unet.output_proj.weight.grad.cpu(), 3.0 * g["eval_grad.unet.output_proj.weight"], GRAD_RTOL, atol=1e-8)
    with torch.no_grad():
        loss = m(dev(g["x"]), dev(g["cond"]), **args)       # no graph, forward only
    assert not loss.requires_grad
    assert_close(loss.item(), g["eval_loss"], 1e-5)


def test_full_shape_grads_vs_oracle():
    shapes = O.param_shapes(50, 1900, 50, 3, FULL_H, 128)
    sd = O.init_state_dict(shapes, seed=5)
    gen = torch.Generator().manual_seed(8)
    for k in sd:                       # non-trivial GroupNorm affine
        if k.endswith((".1.weight", ".5.weight")):
            sd[k] = 1 + 0.2 * torch.randn(sd[k].shape, generator=gen)
        if k.endswith((".1.bias", ".5.bias")):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=gen)
    m = BiologyAwareDiffusionModel(config=config(FULL_H), **FULL)
    m.load_state_dict(sd, strict=False)
    m = m.cuda().train()
    B = 192
    x = torch.randn(B, 2000, generator=gen)
    x[:, :50] = (x[:, :50] > 0).float()
    cond = torch.randn(B, 3, generator=gen)
    t = torch.randint(0, 1000, (B,), generator=gen)
    noise = torch.randn(B, 2000, generator=gen)
    widths = [512, 256, 256, 512, 256]
    masks = [(torch.rand(B, w, generator=gen) >= 0.2).float() for w in widths]
    bufs = O.schedule_buffers("cosine", 1000)
    ref_loss, ref_grads = O.training_loss_and_grads(sd, bufs, x, cond, t, noise, 3, 128, masks, 0.2)
    loss = m(x.cuda(), cond.cuda(), t=t.cuda(), noise=noise.cuda(), dropout_masks=[k.cuda() for k in masks])
    loss.backward()
    assert_close(loss.item(), ref_loss, 1e-5, what="loss")
    for k, p in m.named_parameters():
        assert_close(p.grad.cpu(), ref_grads[k], GRAD_RTOL, atol=1e-9, what=f"grad {k}")


def test_philox_dropout_grads_vs_oracle(golden_dir):
    """Train mode with in-kernel Philox masks at a multi-tile batch: the oracle, fed the host
    restatement of those masks, reproduces loss and gradients (backward regenerates the masks)."""
    from helpers import philox_keep_mask
    m = small_model(golden_dir).train()
    sd = {k: v for k, v in golden_small_sd(golden_dir).items() if k.startswith(("condition_embed", "unet"))}
    n, seed = 512, 77
    x = torch.randn(n, 40, generator=torch.Generator().manual_seed(0))
    cond = torch.randn(n, 3, generator=torch.Generator().manual_seed(1))
    t = torch.randint(0, 1000, (n,), generator=torch.Generator().manual_seed(2))
    noise = torch.randn(n, 40, generator=torch.Generator().manual_seed(3))
    loss = m(x.cuda(), cond.cuda(), t=t.cuda(), noise=noise.cuda(), seed=seed)
    loss.backward()
    again = m(x.cuda(), cond.cuda(), t=t.cuda(), noise=noise.cuda(), seed=seed).item()
    assert abs(loss.item() - again) < 1e-6 * abs(again)      # same seed, same masks (the loss sum uses float atomics)
    loss_eval = small_model(golden_dir)(x.cuda(), cond.cuda(), t=t.cuda(), noise=noise.cuda())
    assert abs(loss.item() - loss_eval.item()) > 1e-4                                                # dropout is active
    masks = [torch.from_numpy(philox_keep_mask(seed, n, w, b, 0.2)) for b, w in enumerate([64, 32, 32, 64, 32])]
    bufs = {k: v.cpu() for k, v in m.state_dict().items() if "alpha" in k or k == "betas"}
    ref_loss, ref_grads = O.training_loss_and_grads(sd, bufs, x, cond, t, noise, 3, 128, masks, 0.2)
    assert_close(loss.item(), ref_loss, 1e-5, what="loss")
    for k, p in m.named_parameters():
        assert_close(p.grad.cpu(), ref_grads[k], GRAD_RTOL, atol=1e-8, what=f"grad {k}")


def test_philox_dropout_equals_injected_masks(golden_dir):
    """The in-kernel Philox keep-masks are exactly the masks a host restatement of Philox4x32-10
    produces; with those injected, loss and every gradient agree with the Philox run."""
    from helpers import philox_keep_mask
    m = small_model(golden_dir).train()
    n = 300
    x = torch.randn(n, 40, generator=torch.Generator().manual_seed(0)).cuda()
    cond = torch.randn(n, 3, generator=torch.Generator().manual_seed(1)).cuda()
    t = torch.randint(0, 1000, (n,), generator=torch.Generator().manual_seed(2)).cuda()
    noise = torch.randn(n, 40, generator=torch.Generator().manual_seed(3)).cuda()
    seed = (5 << 40) + 77
    loss_a = m(x, cond, t=t, noise=noise, seed=seed)
    loss_a.backward()
    ga = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    widths = [64, 32, 32, 64, 32]
    masks = [torch.from_numpy(philox_keep_mask(seed, n, w, b, 0.2)).cuda() for b, w in enumerate(widths)]
    assert 0.7 < float(masks[0].mean()) < 0.9
    loss_b = m(x, cond, t=t, noise=noise, dropout_masks=masks)
    loss_b.backward()
    assert_close(loss_a.item(), loss_b.item(), 1e-6, what="loss philox vs injected")
    for k, p in m.named_parameters():
        assert_close(ga[k].cpu(), p.grad.cpu(), 1e-5, atol=1e-9, what=f"grad {k}")


def test_mixup_kernel(golden_dir):
    g = load_golden(golden_dir, "g6_train_epoch")
    m = small_model(golden_dir)
    data, cond, surv = dev(g["data"][:16]), dev(g["cond"][:16]), dev(g["surv"][:16])
    lam, perm = float(g["lam"][0]), torch.from_numpy(g["perm"][0])
    eng = m._engine()
    from osteosarcoma_diffusionmodel_amd import _lib as L
    od, oc, os_ = torch.empty_like(data), torch.empty_like(cond), torch.empty_like(surv)
    perm_d = perm.cuda()
    L.check(L.lib().osd_mixup(eng.handle, L.ptr(data), L.ptr(cond), L.ptr(surv), L.ptr(perm_d), lam, 16,
                              L.ptr(od), L.ptr(oc), L.ptr(os_)))
    rd, rc, rs = O.mixup(data.cpu(), cond.cpu(), surv.cpu(), lam, perm)
    assert np.array_equal(od.cpu().numpy(), rd.numpy())
    assert np.array_equal(oc.cpu().numpy(), rc.numpy())
    assert np.array_equal(os_.cpu().numpy(), rs.numpy())
    with pytest.raises(RuntimeError):
        MixupAugmentation(0.2)({"data": data.cpu(), "conditions": cond.cpu(), "survival": surv.cpu()})


def test_fused_clip_adamw_vs_torch():
    gen = torch.Generator().manual_seed(0)
    n = 100003
    p0 = torch.randn(n, generator=gen)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=1e-2)
    conf = config(SM_H, p=0.0)
    m = BiologyAwareDiffusionModel(config=conf, **SM).cuda()
    eng = m._engine()
    from osteosarcoma_diffusionmodel_amd import _lib as L
    p = p0.clone().cuda()
    mm, vv = torch.zeros_like(p), torch.zeros_like(p)
    norm = torch.zeros(1, device="cuda")
    for step in range(1, 6):
        gr = torch.randn(n, generator=gen) * (10.0 if step % 2 else 0.001)      # clipped and unclipped steps
        ref.grad = gr.clone()
        tn = torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        gd = gr.cuda()
        L.check(L.lib().osd_clip_adamw_step(eng.handle, L.ptr(p), L.ptr(gd), L.ptr(mm), L.ptr(vv), n, 1e-3, 0.9, 0.999, 1e-8,
                                            1e-2, 1.0, step, L.ptr(norm)))
        assert_close(norm.item(), tn.item(), 1e-6)
        assert_close(gd.cpu(), ref.grad, 1e-6, what="clipped grad")
        assert_close(p.cpu(), ref.detach(), 1e-6, what=f"param step {step}")
    st = opt.state[ref]
    assert_close(mm.cpu(), st["exp_avg"], 1e-6)
    assert_close(vv.cpu(), st["exp_avg_sq"], 5e-6)


def test_trainer_epoch_vs_reference(golden_dir, tmp_path):
    """utils/train.py Trainer.train_epoch on 64 rows (4 steps: mixup, loss, backward, clip, AdamW)
    with the reference's recorded lam / perm / t / noise."""
    g = load_golden(golden_dir, "g6_train_epoch")
    conf = config(SM_H, p=0.0)
    conf["training"] = {"learning_rate": 1e-4, "weight_decay": 1e-5, "patience": 100, "min_delta": 1e-4,
                        "augmentation": {"mixup_alpha": 0.2}, "save_dir": str(tmp_path), "num_epochs": 1,
                        "save_frequency": 10, "val_split": 0.2, "random_seed": 42, "batch_size": 16}
    m = BiologyAwareDiffusionModel(config=conf, **SM)
    m.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd0.")})
    tr = Trainer(m, [], [], conf, device="cuda")
    m.train()
    eng = m._engine()
    from osteosarcoma_diffusionmodel_amd import _lib as L
    losses = []
    for it in range(4):
        sl = slice(16 * it, 16 * it + 16)
        data, cond, surv = dev(g["data"][sl]), dev(g["cond"][sl]), dev(g["surv"][sl])
        od, oc, os_ = torch.empty_like(data), torch.empty_like(cond), torch.empty_like(surv)
        perm_d = dev(g["perm"][it])
        L.check(L.lib().osd_mixup(eng.handle, L.ptr(data), L.ptr(cond), L.ptr(surv), L.ptr(perm_d), float(g["lam"][it]),
                                  16, L.ptr(od), L.ptr(oc), L.ptr(os_)))
        losses.append(tr.train_step(od, oc, t=dev(g["t"][it]), noise=dev(g["noise"][it])).item())
    assert_close(np.mean(losses), g["avg_loss"], 1e-5, what="epoch loss")
    sd1 = m.state_dict()
    for k, _ in m.named_parameters():
        assert_close(sd1[k].cpu(), g["sd1." + k], 2e-5, atol=1e-8, what=f"param {k}")
    osd = tr.optimizer.state_dict()
    names = [k for k, _ in m.named_parameters()]
    for i, k in enumerate(names):
        assert_close(osd["state"][i]["exp_avg"].cpu(), g["exp_avg." + k], 1e-4, atol=1e-10, what=f"exp_avg {k}")
        assert_close(osd["state"][i]["exp_avg_sq"].cpu(), g["exp_avg_sq." + k], 1e-4, atol=1e-14, what=f"exp_avg_sq {k}")
        assert float(osd["state"][i]["step"]) == 4.0
    # the optimizer state loads into a stock torch AdamW (checkpoint interchange)
    stock = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5)
    stock.load_state_dict(osd)
    # weights changed behind autograd's back: inference must see them
    eps = m.eval().predict_noise(dev(g["data"][:4]), 10, dev(g["cond"][:4]))
    sdp = {k: v.cpu() for k, v in sd1.items() if k.startswith(("condition_embed", "unet"))}
    ref = O.unet_forward(sdp, torch.from_numpy(g["data"][:4]), torch.full((4,), 10 / 1000), O.condition_embed(sdp, torch.from_numpy(g["cond"][:4])), 3, 128)
    assert_close(eps.cpu(), ref, 1e-5, what="post-training inference")


def test_trainer_full_loop_runs(golden_dir, tmp_path):
    """Trainer.train(): epochs, validation, ReduceLROnPlateau, checkpoint files and keys, early stop."""
    conf = config(SM_H, p=0.2)
    conf["training"] = {"learning_rate": 1e-3, "weight_decay": 1e-5, "patience": 2, "min_delta": 10.0,
                        "augmentation": {"mixup_alpha": 0.2}, "save_dir": str(tmp_path), "num_epochs": 6,
                        "save_frequency": 2, "val_split": 0.2, "random_seed": 42, "batch_size": 32}
    gen = torch.Generator().manual_seed(0)
    n = 256
    rows = [{"data": torch.randn(40, generator=gen), "conditions": torch.randn(3, generator=gen), "survival": torch.rand(1, generator=gen)[0]}
            for _ in range(n)]
    loader = torch.utils.data.DataLoader(rows, batch_size=32, drop_last=True)
    m = BiologyAwareDiffusionModel(config=conf, **SM)
    tr = Trainer(m, loader, loader, conf, device="cuda")
    hist = tr.train()
    assert len(hist["train_loss"]) == len(hist["val_loss"]) == 3          # min_delta=10 -> stop after patience+1 epochs
    assert all(np.isfinite(hist["train_loss"])) and all(np.isfinite(hist["val_loss"]))
    ck = torch.load(tmp_path / "best_model.pt", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "val_loss", "config"}
    assert list(ck["model_state_dict"]) == list(m.state_dict())
    assert (tmp_path / "checkpoint_epoch_0.pt").exists()
    assert hist["train_loss"][-1] < hist["train_loss"][0] + 0.5


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_custom_loss_through_differentiable_predict_noise(golden_dir, mode):
    """osd_denoiser_forward_train + osd_denoiser_backward: a Huber loss on eps_hat (config.yaml:47 lists l1 / l2 / huber)
    back-propagated to every parameter and to x_t, against autograd over the oracle."""
    g = load_golden(golden_dir, "g3g4_small_model")
    sd = golden_small_sd(golden_dir)
    params = {k: v for k, v in sd.items() if k.startswith(("condition_embed", "unet"))}
    m = small_model(golden_dir)
    gen = torch.Generator().manual_seed(3)
    rows = 37
    x_t = torch.randn(rows, 40, generator=gen)
    cond = torch.randn(rows, 3, generator=gen)
    t = torch.randint(0, 1000, (rows,), generator=gen)
    target = torch.randn(rows, 40, generator=gen)
    masks = None
    if mode == "train":
        m.train()
        masks = [(torch.rand(rows, c, generator=gen) > 0.2).float() for c in (SM_H[1], SM_H[2], SM_H[2], SM_H[1], SM_H[0])]
    # oracle
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xo = x_t.clone().requires_grad_(True)
    pred = O.unet_forward(leaves, xo, t.float() / 1000, O.condition_embed(leaves, cond), len(SM_H), 128, masks, 0.2 if masks else 0.0)
    ref = torch.nn.functional.smooth_l1_loss(pred, target)
    gref = torch.autograd.grad(ref, [xo] + list(leaves.values()))
    # device
    xd = x_t.cuda().requires_grad_(True)
    eps = m.predict_noise(xd, t.cuda(), cond.cuda(), dropout_masks=[k.cuda() for k in masks] if masks else None)
    assert eps.requires_grad
    loss = torch.nn.functional.smooth_l1_loss(eps, target.cuda())
    loss.backward()
    assert_close(eps.detach().cpu(), pred.detach(), 1e-5, what="eps_hat")
    assert_close(loss.item(), ref.item(), 1e-5, what="huber loss")
    assert_close(xd.grad.cpu(), gref[0], GRAD_RTOL, atol=1e-9, what="dL/dx_t")
    named = dict(m.named_parameters())
    for k, gr in zip(leaves, gref[1:]):
        assert_close(named[k].grad.cpu(), gr, GRAD_RTOL, atol=1e-9, what=f"grad {k}")
    # a second forward invalidates the first one's workspace: its backward must refuse, not compute garbage
    e1 = m.predict_noise(xd, t.cuda(), cond.cuda(), dropout_masks=[k.cuda() for k in masks] if masks else None)
    m.predict_noise(xd, t.cuda(), cond.cuda(), dropout_masks=[k.cuda() for k in masks] if masks else None)
    with pytest.raises(RuntimeError):
        e1.sum().backward()


@pytest.mark.parametrize("cond_dim", [8, 12, 6, 4, 1])
@pytest.mark.parametrize("train_streams", [1, 2])
def test_wide_condition_grads_vs_oracle(cond_dim, train_streams):
    """condition_dim 8 / 12: the ConditionalEmbedding's first weight gradient (64 x cond_dim) is neither the small kernel's
    (cond_dim <= 4) nor the grouped launch's (k_in >= 16) and takes the immediate split-K path on the side stream, which shares
    the slab workspace with the end-of-pass grouped launch on the main stream (round-2 advisor finding: the main stream must
    wait for it).  cond_dim 6 takes the guarded path without slabs.  Multi-slice batch so that the slabs are really used.
    cond_dim 4 / 1: the widest and the narrowest first Linear whose weight gradient rides inside k_cond_bwd (csrc/k_train.hip:
    16 partial copies folded by the last workgroup), here at hidden_dims[0] = 64 -- two of the eight K groups of its first GEMM."""
    dims = dict(mutation_dim=8, expression_dim=48, pathway_dim=8, condition_dim=cond_dim)
    H = [64, 128, 64]
    shapes = O.param_shapes(8, 48, 8, cond_dim, H, 128)
    sd = O.init_state_dict(shapes, seed=3)
    m = BiologyAwareDiffusionModel(config=config(H), **dims)
    m.load_state_dict(sd, strict=False)
    m = m.cuda().train()
    m.train_streams = train_streams
    gen = torch.Generator().manual_seed(cond_dim)
    Bn = 1024
    x = torch.randn(Bn, 64, generator=gen)
    cond = torch.randn(Bn, cond_dim, generator=gen)
    t = torch.randint(0, 1000, (Bn,), generator=gen)
    noise = torch.randn(Bn, 64, generator=gen)
    masks = [(torch.rand(Bn, w, generator=gen) >= 0.2).float() for w in [128, 64, 64, 128, 64]]
    bufs = O.schedule_buffers("cosine", 1000)
    ref_loss, ref_grads = O.training_loss_and_grads(sd, bufs, x, cond, t, noise, 3, 128, masks, 0.2)
    for rep in range(3):               # a race shows as a flaky tensor: repeat the pass
        m.zero_grad()
        loss = m(x.cuda(), cond.cuda(), t=t.cuda(), noise=noise.cuda(), dropout_masks=[k.cuda() for k in masks])
        loss.backward()
        assert_close(loss.item(), ref_loss, 1e-5, what="loss")
        for k, p in m.named_parameters():
            assert_close(p.grad.cpu(), ref_grads[k], GRAD_RTOL, atol=1e-8, what=f"grad {k} (rep {rep})")


@pytest.mark.parametrize("resident", [True, False])
def test_train_epoch_paths_give_the_reference_epoch(golden_dir, tmp_path, resident):
    """Trainer.train_epoch over a DataLoader of fixture g6's 64 rows -- once replayed from HBM (ResidentSplit: rows gathered,
    mixed up and noised by one kernel, osd_train_batch_source), once iterating the DataLoader as the reference does
    (utils/train.py:204-250) -- with the reference's recorded lam / perm / t / noise injected: both must land on the
    reference's post-epoch parameters, and the two paths on each other's."""
    from osteosarcoma_diffusionmodel_amd.train import OsteosarcomaDataset
    g = load_golden(golden_dir, "g6_train_epoch")
    conf = config(SM_H, p=0.0)
    conf["training"] = {"learning_rate": 1e-4, "weight_decay": 1e-5, "patience": 100, "min_delta": 1e-4,
                        "augmentation": {"mixup_alpha": 0.2}, "save_dir": str(tmp_path), "num_epochs": 1,
                        "save_frequency": 10, "val_split": 0.2, "random_seed": 42, "batch_size": 16, "resident_dataset": resident}
    ds = object.__new__(OsteosarcomaDataset)
    ds.data, ds.conditions, ds.survival_days = (torch.from_numpy(g[k]).float() for k in ("data", "cond", "surv"))
    loader = torch.utils.data.DataLoader(ds, batch_size=16, shuffle=False, num_workers=0, drop_last=True)
    m = BiologyAwareDiffusionModel(config=conf, **SM)
    m.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd0.")})
    tr = Trainer(m, loader, loader, conf, device="cuda")
    step = {"i": 0}
    tr.mixup.draw = lambda n, device: (float(g["lam"][step["i"]]), dev(g["perm"][step["i"]]))              # per step (DataLoader path)
    tr.mixup.draw_epoch = lambda sizes, device, seed_fn=None: ([float(v) for v in g["lam"][:len(sizes)]],     # per epoch (resident path)
                                                               [dev(g["perm"][i]) for i in range(len(sizes))], [None] * len(sizes))
    orig = tr.train_step

    def injected(*a, **k):
        it = step["i"]
        out = orig(*a, t=dev(g["t"][it]), noise=dev(g["noise"][it]), **k)
        step["i"] += 1
        return out

    tr.train_step = injected
    avg = tr.train_epoch()
    assert step["i"] == 4 and bool(tr.resident) == resident
    assert_close(avg, g["avg_loss"], 1e-5, what="epoch loss")
    sd1 = m.state_dict()
    for k, _ in m.named_parameters():
        assert_close(sd1[k].cpu(), g["sd1." + k], 2e-5, atol=1e-8, what=f"param {k}")
    val = tr.validate()                       # the resident validation pass runs (4 batches of the same rows)
    assert np.isfinite(val)


def test_fallback_groupnorm_backward_is_deterministic_and_trainer_follows_option_changes(tmp_path):
    """(i) Block widths other than 256 / 512 (here 64 / 128: GroupNorm groups of 8 / 16 channels) take the stand-alone GroupNorm
    backward kernel; its dgamma / dbeta / dbias come from a fixed-order partial reduce, so two passes over the same inputs agree bit
    for bit (round-2 advisor finding: they were float atomics).  (ii) A tunable changed on the model after the Trainer was built
    (here train_streams) reaches the Trainer's fast path."""
    H = [64, 128, 64]
    dims = dict(mutation_dim=8, expression_dim=48, pathway_dim=8, condition_dim=3)
    conf = config(H)
    conf["training"] = {"learning_rate": 1e-4, "weight_decay": 1e-5, "patience": 10, "min_delta": 1e-4, "augmentation": {"mixup_alpha": 0.0},
                        "save_dir": str(tmp_path), "num_epochs": 1, "save_frequency": 10, "val_split": 0.2, "random_seed": 42, "batch_size": 512}
    torch.manual_seed(3)
    m = BiologyAwareDiffusionModel(config=conf, **dims).cuda().train()
    gen = torch.Generator().manual_seed(1)
    x, c = torch.randn(512, 64, generator=gen).cuda(), torch.randn(512, 3, generator=gen).cuda()
    t, nz = torch.randint(0, 1000, (512,), generator=gen).cuda(), torch.randn(512, 64, generator=gen).cuda()
    grads = []
    for _ in range(2):
        m.zero_grad()
        m(x, c, t=t, noise=nz, seed=5).backward()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters()
                      if k.startswith(("unet.encoder", "unet.decoder", "unet.bottleneck")) and
                      k.endswith((".1.weight", ".1.bias", ".5.weight", ".5.bias", ".0.bias", ".4.bias"))})
    assert grads[0] and all(torch.equal(grads[0][k], grads[1][k]) for k in grads[0])
    tr = Trainer(m, [], [], conf, device="cuda")
    tr.train_step(x, c, t=t, noise=nz, seed=5)
    import ctypes as C
    from osteosarcoma_diffusionmodel_amd import _lib as L
    v = C.c_int64()
    m.train_streams = 1
    tr.train_step(x, c, t=t, noise=nz, seed=5)
    L.check(L.lib().osd_get_option(tr._engine.handle, b"train_streams", C.byref(v)))
    assert v.value == 1
    m.train_streams = 2
    tr.train_step(x, c, t=t, noise=nz, seed=5)
    L.check(L.lib().osd_get_option(tr._engine.handle, b"train_streams", C.byref(v)))
    assert v.value == 2


def test_batch_source_with_constraint_losses(golden_dir):
    """Rows from the resident dataset (osd_train_batch_source) while the constraint losses are configured: they read the mixed x0,
    which the fused gather + mixup + q_sample kernel then also materialises -- loss and gradients must equal the call that is
    handed the same mixed rows as tensors."""
    m = small_model(golden_dir).train()
    m.set_constraints(pathways=[[8, 9, 10, 11], [12, 13, 14]], mutation_columns=[0, 1], target_columns=[32, 33], pathway_weight=0.5, mutexpr_weight=0.25)
    gen = torch.Generator().manual_seed(7)
    N, B = 200, 64
    data, cond = torch.randn(N, 40, generator=gen).cuda(), torch.randn(N, 3, generator=gen).cuda()
    idx = torch.randperm(N, generator=gen)[:B].cuda()
    perm = torch.randperm(B, generator=gen).cuda()
    lam = 0.3
    t, nz = torch.randint(0, 1000, (B,), generator=gen).cuda(), torch.randn(B, 40, generator=gen).cuda()
    from osteosarcoma_diffusionmodel_amd.train import _loss_fwd_bwd, MixupAugmentation
    mix = MixupAugmentation(0.2).mix({"data": data[idx], "conditions": cond[idx], "survival": torch.zeros(B, device="cuda")}, lam, perm)
    ga = [torch.empty_like(p) for p in m.parameters()]
    gb = [torch.empty_like(p) for p in m.parameters()]
    la = _loss_fwd_bwd(m, mix["data"], mix["conditions"], L_ptrs(ga), t=t, noise=nz, seed=9)
    lb = _loss_fwd_bwd(m, None, None, L_ptrs(gb), t=t, noise=nz, seed=9, source=(data, cond, idx, idx[perm], lam))
    assert_close(lb.item(), la.item(), 1e-6, what="loss")
    for (k, _), a, b in zip(m.named_parameters(), ga, gb):
        assert_close(b.cpu(), a.cpu(), 1e-5, atol=1e-9, what=f"grad {k}")


def L_ptrs(tensors):
    from osteosarcoma_diffusionmodel_amd import _lib as L
    return L.ptr_array(tensors)
