"""GPU: BASELINE config 2 at its stated batch (D = 2000, B = 4096, dropout 0.2) against the CPU oracle, plus the
full-shape trained-weights chain of SURVEY section 8d.

At B = 4096 the library takes dispatch branches no smaller batch reaches (tile selection in csrc/launch.h, split-K
slice count and slab reduction in csrc/train.hip, the two-stream backward), so loss, all 52 gradients and the
parameters / AdamW moments after one ``Trainer.train_step`` are compared here with the one- and the two-stream
backward, with injected keep-masks and with the in-kernel Philox masks (the oracle is fed the host restatement of
those).  Stated fp32 tolerances: loss 1e-5 relative; each gradient tensor max|d| <= 5e-5 * max|ref|; parameters after
the step 2e-5 * max|ref| plus the gradient tolerance propagated through the first AdamW update (steep where the
clipped gradient is comparable to eps = 1e-8); first moments 1e-4, second moments 2e-4 relative.  References: models/diffusion.py:344-380,
utils/train.py:236-244."""
import copy

import numpy as np
import pytest
import torch

from oracle import diffusion_oracle as O
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
from osteosarcoma_diffusionmodel_amd.train import Trainer
from helpers import FULL, FULL_H, assert_close, config, philox_keep_mask

pytestmark = pytest.mark.gpu
GRAD_RTOL = 5e-5
B = 4096
WIDTHS = [512, 256, 256, 512, 256]
SEED = (3 << 33) + 4242
LR, WD = 1e-4, 1e-5

_cache = {}


def _inputs():
    if "in" not in _cache:
        shapes = O.param_shapes(50, 1900, 50, 3, FULL_H, 128)
        sd = O.init_state_dict(shapes, seed=11)
        gen = torch.Generator().manual_seed(21)
        for k in sd:                       # non-trivial GroupNorm affine
            if k.endswith((".1.weight", ".5.weight")):
                sd[k] = 1 + 0.2 * torch.randn(sd[k].shape, generator=gen)
            if k.endswith((".1.bias", ".5.bias")):
                sd[k] = 0.1 * torch.randn(sd[k].shape, generator=gen)
        x = torch.randn(B, 2000, generator=gen)
        x[:, :50] = (x[:, :50] > 0).float()
        cond = torch.randn(B, 3, generator=gen)
        t = torch.randint(0, 1000, (B,), generator=gen)
        noise = torch.randn(B, 2000, generator=gen)
        injected = [(torch.rand(B, w, generator=gen) >= 0.2).float() for w in WIDTHS]
        _cache["in"] = (sd, x, cond, t, noise, injected)
    return _cache["in"]


def _oracle(mask_mode):
    """loss, gradients and the post-step parameters / moments of the oracle for one mask mode (cached: ~5 s each)."""
    key = ("ref", mask_mode)
    if key not in _cache:
        sd, x, cond, t, noise, injected = _inputs()
        if mask_mode == "injected":
            masks = injected
        else:
            masks = [torch.from_numpy(philox_keep_mask(SEED, B, w, b, 0.2)) for b, w in enumerate(WIDTHS)]
        bufs = O.schedule_buffers("cosine", 1000)
        loss, grads = O.training_loss_and_grads(sd, bufs, x, cond, t, noise, 3, 128, masks, 0.2)
        names = list(sd)
        clipped, norm = O.clip_grad_norm([grads[k] for k in names], 1.0)
        p1 = [sd[k].clone() for k in names]
        m1 = [torch.zeros_like(sd[k]) for k in names]
        v1 = [torch.zeros_like(sd[k]) for k in names]
        O.adamw_step(p1, clipped, m1, v1, 1, lr=LR, weight_decay=WD)
        _cache[key] = dict(loss=float(loss), grads=grads, norm=float(norm), names=names, p1=dict(zip(names, p1)), gclip=dict(zip(names, clipped)),
                           m1=dict(zip(names, m1)), v1=dict(zip(names, v1)))
    return _cache[key]


def _assert_params_close(got, want, gclip, name):
    """The first AdamW step moves an element by lr * g / (|g| + eps) (m_hat = g, sqrt(v_hat) = |g|): +-lr for all but the
    elements whose clipped gradient is comparable to eps = 1e-8, where the update is steep in g.  The stated gradient
    tolerance (5e-5 * max|g| per tensor) is therefore propagated through that map: an element may deviate by
    2e-5 * max|p| + |d step / d g| * 5e-5 * max|g|, never by more than the 2 lr of a flipped sign."""
    got, want, g = got.double().numpy(), want.double().numpy(), np.abs(gclip.double().numpy())
    d = np.abs(got - want)
    assert np.isfinite(got).all(), name
    eps = 1e-8
    sens = LR * eps / (g + eps) ** 2
    allowed = 2e-5 * np.abs(want).max() + 1e-8 + np.minimum(sens * GRAD_RTOL * g.max(), 2.0 * LR)
    worst = (d - allowed).max()
    assert worst <= 0, f"param {name}: an element exceeds its propagated tolerance by {worst:.3e} (max|d|={d.max():.3e})"


def _model(train_streams, precision=None):
    sd = _inputs()[0]
    conf = config(FULL_H)
    conf["training"] = {"learning_rate": LR, "weight_decay": WD, "patience": 10, "min_delta": 1e-4,
                        "augmentation": {"mixup_alpha": 0.0}, "save_dir": "/tmp/osd_cfg2", "num_epochs": 1,
                        "save_frequency": 10, "val_split": 0.2, "random_seed": 42, "batch_size": B}
    m = BiologyAwareDiffusionModel(config=conf, **FULL)
    m.load_state_dict(sd, strict=False)
    m = m.cuda().train()
    m.train_streams = train_streams
    m.precision = precision
    return m, conf


@pytest.mark.parametrize("mask_mode", ["injected", "philox"])
@pytest.mark.parametrize("train_streams", [1, 2])
def test_config2_batch_grads_and_step_vs_oracle(train_streams, mask_mode, tmp_path):
    sd, x, cond, t, noise, injected = _inputs()
    ref = _oracle(mask_mode)
    kw = dict(t=t.cuda(), noise=noise.cuda())
    if mask_mode == "injected":
        kw["dropout_masks"] = [k.cuda() for k in injected]
    else:
        kw["seed"] = SEED
    # (i) loss and the 52 gradients through the autograd entry (loss.backward())
    m, conf = _model(train_streams)
    loss = m(x.cuda(), cond.cuda(), **kw)
    loss.backward()
    assert_close(loss.item(), ref["loss"], 1e-5, what="loss")
    for k, p in m.named_parameters():
        assert_close(p.grad.cpu(), ref["grads"][k], GRAD_RTOL, atol=1e-9, what=f"grad {k}")
    # (ii) one Trainer.train_step: flat gradients -> fused clip + AdamW
    m2, conf = _model(train_streams)
    conf["training"]["save_dir"] = str(tmp_path)
    tr = Trainer(m2, [], [], conf, device="cuda")
    loss2 = tr.train_step(x.cuda(), cond.cuda(), **kw)
    assert_close(loss2.item(), ref["loss"], 1e-5, what="train_step loss")
    assert_close(tr.optimizer.grad_norm.item(), ref["norm"], 2e-5, what="pre-clip gradient norm")
    osd = tr.optimizer.state_dict()
    for i, (k, p) in enumerate(m2.named_parameters()):
        _assert_params_close(p.detach().cpu(), ref["p1"][k], ref["gclip"][k], k)
        assert_close(osd["state"][i]["exp_avg"].cpu(), ref["m1"][k], 1e-4, atol=1e-12, what=f"exp_avg {k}")
        assert_close(osd["state"][i]["exp_avg_sq"].cpu(), ref["v1"][k], 2e-4, atol=1e-16, what=f"exp_avg_sq {k}")
        assert float(osd["state"][i]["step"]) == 1.0


@pytest.mark.parametrize("mask_mode", ["injected", "philox"])
def test_config2_weight_gradients_on_the_bf16_pipe_vs_oracle(mask_mode):
    """``precision = "bf16x3"`` in training: every weight gradient of the step (the grouped launch, csrc/wgrad_group.h: both operands
    split into three bf16 planes as they are staged, six bf16 MFMAs per product, fp32 accumulation) -- loss and all 52 gradients
    against the oracle at the UNCHANGED config-2 tolerances, twice (cached work list).  Forward and the dgrad chain stay fp32."""
    sd, x, cond, t, noise, injected = _inputs()
    ref = _oracle(mask_mode)
    kw = dict(t=t.cuda(), noise=noise.cuda())
    if mask_mode == "injected":
        kw["dropout_masks"] = [k.cuda() for k in injected]
    else:
        kw["seed"] = SEED
    m, _ = _model(2, precision="bf16x3")
    for rep in range(2):
        m.zero_grad()
        loss = m(x.cuda(), cond.cuda(), **kw)
        loss.backward()
        assert_close(loss.item(), ref["loss"], 1e-5, what="loss")
        for k, p in m.named_parameters():
            assert_close(p.grad.cpu(), ref["grads"][k], GRAD_RTOL, atol=1e-9, what=f"grad {k} (bf16x3 weight gradients, pass {rep})")


@pytest.mark.parametrize("n", [4096, 2100])
def test_config2_forward_trunk_as_one_squad_launch_vs_per_layer_launches(n):
    """csrc/train_squad.h: from 2 048 rows on the ten Linear+GroupNorm+SiLU(+Dropout) layers of a training forward pass run as ONE
    launch of squads (eight workgroups per 64 patients; two K-halves per output: another fp32 summation order).  The test above
    holds it to the oracle at B = 4096; here it is compared with the per-layer launches it replaces (``train_squad = False``) on
    the same inputs and the same Philox dropout draws: loss and all 52 gradients, at the full batch and at 2 100 rows (a last panel
    of 52 patients, below one workgroup per CU)."""
    sd, x, cond, t, noise, _ = _inputs()
    kw = dict(t=t[:n].cuda(), noise=noise[:n].cuda(), seed=SEED)
    out = {}
    for squad in (False, True, 2):
        m, _ = _model(2)
        m.train_squad = squad
        loss = m(x[:n].cuda(), cond[:n].cuda(), **kw)
        loss.backward()
        out[squad] = (loss.item(), {k: p.grad.detach().cpu() for k, p in m.named_parameters()})
    assert_close(out[True][0], out[False][0], 2e-6, what="loss, squad forward vs per-layer launches")
    for k in out[True][1]:
        assert_close(out[True][1][k], out[False][1][k], GRAD_RTOL, atol=1e-9, what=f"grad {k}, squad forward vs per-layer launches")
    # it really is another summation order (bit-equal gradients everywhere would mean the switch did nothing)
    assert any(not torch.equal(out[True][1][k], out[False][1][k]) for k in out[True][1])
    # ... and the dgrad chain as one launch of squads as well (csrc/train_squad_bwd.h; single-GPU steps)
    assert_close(out[2][0], out[False][0], 2e-6, what="loss, squad forward + backward vs per-layer launches")
    for k in out[2][1]:
        assert_close(out[2][1][k], out[False][1][k], GRAD_RTOL, atol=1e-9, what=f"grad {k}, squad forward + backward vs per-layer launches")
    assert any(not torch.equal(out[2][1][k], out[True][1][k]) for k in out[2][1])


@pytest.mark.parametrize("n", [4096, 2100, 37])
def test_config2_conditioning_branch_backward_as_one_launch_vs_four(n):
    """k_cond_bwd (csrc/k_train.hip): below h0 the backward of the conditioning branch -- the scatter of dL/dh0 into the time table's
    rows, cond_proj's and ConditionalEmbedding's second Linear's dgrads, the SiLU backward -- is one launch of 32-row workgroups
    instead of four dependent ones.  Same inputs, same draws: every gradient against the four-launch path (``cond_bwd_fused =
    False``); the six tensors it feeds (time_proj, cond_proj and the embedding's weights) must not be bit-equal everywhere -- the
    sums run in another order -- and a ragged last workgroup (2 100 = 65 * 32 + 20 rows; 37 rows) is covered."""
    sd, x, cond, t, noise, _ = _inputs()
    kw = dict(t=t[:n].cuda(), noise=noise[:n].cuda(), seed=SEED)
    out = {}
    for fused in (False, True):
        m, _ = _model(2)
        m.cond_bwd_fused = fused
        loss = m(x[:n].cuda(), cond[:n].cuda(), **kw)
        loss.backward()
        out[fused] = (loss.item(), {k: p.grad.detach().cpu() for k, p in m.named_parameters()})
    assert_close(out[True][0], out[False][0], 2e-6, what="loss (the forward pass is the same)")
    for k in out[True][1]:
        assert_close(out[True][1][k], out[False][1][k], GRAD_RTOL, atol=1e-9, what=f"grad {k}, one-launch conditioning backward vs four launches")
    branch = [k for k in out[True][1] if k.startswith("condition_embed") or "time_proj" in k or "cond_proj" in k]
    assert len(branch) >= 6, branch
    assert any(not torch.equal(out[True][1][k], out[False][1][k]) for k in branch)


def test_config2_squad_forward_that_cannot_finish_poisons_the_loss():
    """A training step has no second engine to fall back to inside the call.  A squad whose barrier runs into the spin budget (here:
    one tick) leaves, raises the status word and writes NaN into the loss accumulator: the step's loss is NaN -- visible to
    `Trainer` at its `loss.item()` -- instead of a finite number computed from half-written activations.  With the default
    budget the same call is finite again."""
    sd, x, cond, t, noise, _ = _inputs()
    m, _ = _model(2)
    kw = dict(t=t.cuda(), noise=noise.cuda(), seed=SEED)
    m.chain_spin_budget = 1
    with torch.no_grad():
        bad = m(x.cuda(), cond.cuda(), **kw)
    assert torch.isnan(bad).item()
    m.chain_spin_budget = 500_000_000
    with torch.no_grad():
        good = m(x.cuda(), cond.cuda(), **kw)
    assert torch.isfinite(good).item()


def test_full_shape_trained_weights_chain_vs_oracle():
    """SURVEY section 8d, 'briefly CPU-trained checkpoint' at the BASELINE shape: the oracle trains the D = 2000 model
    for 100 AdamW steps on low-rank structured synthetic patients, then the device runs a T = 200 reverse chain on 32
    rows with those weights and the oracle's draws.  Chain tolerance 5e-5 * max|ref|; the mutation mask must be
    bit-equal wherever the reference is further than that tolerance from the 0.5 threshold
    (models/diffusion.py:382-449, utils/generate.py:135)."""
    T, rows, D = 200, 32, 2000
    shapes = O.param_shapes(50, 1900, 50, 3, FULL_H, 128)
    sd = O.init_state_dict(shapes, seed=3)
    bufs = O.schedule_buffers("cosine", T)
    gen = torch.Generator().manual_seed(77)
    basis = torch.randn(8, D, generator=gen)

    def batch(n):
        x0 = torch.randn(n, 8, generator=gen) @ basis * 0.35 + 0.1 * torch.randn(n, D, generator=gen)
        x0[:, :50] = (x0[:, :50] > 0).float()
        return x0, torch.randn(n, 3, generator=gen)

    names = list(sd)
    m1 = [torch.zeros_like(sd[k]) for k in names]
    m2 = [torch.zeros_like(sd[k]) for k in names]
    first = last = None
    for step in range(1, 101):
        x0, c = batch(96)
        t = torch.randint(0, T, (96,), generator=gen)
        nz = torch.randn(96, D, generator=gen)
        loss, grads = O.training_loss_and_grads(sd, bufs, x0, c, t, nz, 3, 128)
        gl, _ = O.clip_grad_norm([grads[k] for k in names], 1.0)
        O.adamw_step([sd[k] for k in names], gl, m1, m2, step, lr=1e-3, weight_decay=1e-5)
        first = loss.item() if first is None else first
        last = loss.item()
    assert last < 0.95 * first                                   # it learned something
    cond = torch.randn(rows, 3, generator=gen)
    x_T = torch.randn(rows, D, generator=gen)
    zs = torch.randn(T - 1, rows, D, generator=gen)              # draw order t = T-1 .. 1
    ref = O.sample(sd, bufs, cond, x_T, lambda t: zs[T - 1 - t], 3, 128)
    m = BiologyAwareDiffusionModel(config=config(FULL_H, T=T), **FULL)
    m.load_state_dict(sd, strict=False)
    m = m.cuda().eval()
    for graph in (True, False):
        m.use_graph = graph
        out, mask = m.sample(cond.cuda(), rows, x_T=x_T.cuda(), noise=zs.cuda(), return_mutation_mask=True)
        assert_close(out, ref, 5e-5, atol=1e-5, what=f"trained-weights chain (graph={graph})")
        refm = (ref[:, :50] > 0.5).float()
        near = (ref[:, :50] - 0.5).abs() <= 5e-5 * ref.abs().max() + 1e-5
        assert ((mask.cpu() != refm) & ~near).sum().item() == 0
        assert (~near).float().mean().item() > 0.9               # the comparison is not vacuous
