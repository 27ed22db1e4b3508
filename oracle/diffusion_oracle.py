"""CPU oracle for the diffusion hot path -- TEST INFRASTRUCTURE ONLY.

This file is a functional, CPU-only restatement (plain PyTorch CPU ops, fp32 by
default, fp64 on request) of the arithmetic the reference performs on its
diffusion hot path.  It exists to CHECK the HIP path; nothing under
``osteosarcoma_diffusionmodel_amd/`` may import it.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it.

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, produced in the build container by ``tests/golden/make_goldens.py``
(which imports /root/reference) and committed as ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` replays every fixture through this file.

All weights are passed as a plain ``dict[str, Tensor]`` keyed by the
reference's ``state_dict`` names (``condition_embed.mlp.0.weight`` ...,
``unet.encoder.0.0.weight`` ..., see models/diffusion.py:91-256).

Reference lines each function follows are given in its docstring
(paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]

COND_EMBED_WIDTH = 64  # literal in models/diffusion.py:285
GN_GROUPS = 8          # models/diffusion.py:202,206
GN_EPS = 1e-5          # torch.nn.GroupNorm default


# --------------------------------------------------------------------------
# schedule  (models/diffusion.py:298-326)
# --------------------------------------------------------------------------
def beta_schedule(schedule: str, num_steps: int) -> Tensor:
    """betas[T] in fp32, same torch expressions as models/diffusion.py:312-326."""
    if schedule == "linear":
        return torch.linspace(1e-4, 0.02, num_steps)
    if schedule == "cosine":
        s = torch.arange(num_steps + 1, dtype=torch.float32) / num_steps
        abar = torch.cos((s + 0.008) / 1.008 * np.pi / 2) ** 2
        abar = abar / abar[0]
        betas = 1 - (abar[1:] / abar[:-1])
        return torch.clip(betas, 0.0001, 0.9999)
    raise ValueError(f"Unknown schedule: {schedule}")


def schedule_buffers(schedule: str, num_steps: int) -> Dict[str, Tensor]:
    """The four registered buffers (models/diffusion.py:299-310)."""
    betas = beta_schedule(schedule, num_steps)
    abar = torch.cumprod(1.0 - betas, dim=0)
    return {
        "betas": betas,
        "alphas_cumprod": abar,
        "sqrt_alphas_cumprod": torch.sqrt(abar),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - abar),
    }


def posterior_coefficients(bufs: Dict[str, Tensor]) -> Tensor:
    """[T, 6] fp32 table of the 0-d scalars p_sample forms at step t
    (models/diffusion.py:401-419), each evaluated with fp32 tensor ops in the
    order written there:

      col 0  sqrt(1 - abar_t)              col 1  sqrt(abar_t)
      col 2  sqrt(abar_{t-1}) * beta_t     col 3  1 - abar_t
      col 4  sqrt(alpha_t) * (1-abar_{t-1})  col 5  sqrt(var_t)

    Row 0 (t == 0) carries zeros in columns 2, 4, 5: that step returns x0_pred.
    """
    betas, abar = bufs["betas"], bufs["alphas_cumprod"]
    T = betas.shape[0]
    out = torch.zeros(T, 6, dtype=torch.float32)
    for t in range(T):
        alpha_t = 1.0 - betas[t]
        ab = abar[t]
        out[t, 0] = torch.sqrt(1 - ab)
        out[t, 1] = torch.sqrt(ab)
        out[t, 3] = 1 - ab
        if t > 0:
            abp = abar[t - 1]
            out[t, 2] = torch.sqrt(abp) * betas[t]
            out[t, 4] = torch.sqrt(alpha_t) * (1 - abp)
            var = (1 - abp) / (1 - ab) * betas[t]
            out[t, 5] = torch.sqrt(var)
    return out


# --------------------------------------------------------------------------
# embeddings  (models/diffusion.py:91-139)
# --------------------------------------------------------------------------
def time_frequencies(dim: int) -> Tensor:
    """f_k = exp(-k * ln(1e4)/(dim/2 - 1)), fp32 (models/diffusion.py:132-135)."""
    half = dim // 2
    step = np.log(10000) / (half - 1)
    return torch.exp(torch.arange(half) * -step)


def time_embedding(t_norm: Tensor, dim: int) -> Tensor:
    """[sin(t f) | cos(t f)], t already normalised to [0,1) (models/diffusion.py:124-139)."""
    f = time_frequencies(dim).to(t_norm.dtype)
    arg = t_norm[:, None] * f[None, :]
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)


def condition_embed(sd: StateDict, cond: Tensor) -> Tensor:
    """Linear -> SiLU -> Linear (models/diffusion.py:101-105,114)."""
    h = F.linear(cond, sd["condition_embed.mlp.0.weight"], sd["condition_embed.mlp.0.bias"])
    h = F.silu(h)
    return F.linear(h, sd["condition_embed.mlp.2.weight"], sd["condition_embed.mlp.2.bias"])


# --------------------------------------------------------------------------
# denoiser  (models/diffusion.py:198-256)
# --------------------------------------------------------------------------
def block_names(n_hidden: int) -> List[str]:
    """Block prefixes in execution order for len(hidden_dims) == n_hidden."""
    names = [f"unet.encoder.{i}" for i in range(n_hidden - 1)]
    names.append("unet.bottleneck")
    names += [f"unet.decoder.{i}" for i in range(n_hidden - 1)]
    return names


def _half_block(sd: StateDict, prefix: str, lin: int, gn: int, x: Tensor) -> Tensor:
    w, b = sd[f"{prefix}.{lin}.weight"], sd[f"{prefix}.{lin}.bias"]
    g, be = sd[f"{prefix}.{gn}.weight"], sd[f"{prefix}.{gn}.bias"]
    z = F.linear(x, w, b)
    z = F.group_norm(z, GN_GROUPS, g, be, GN_EPS)
    return F.silu(z)


def block_forward(sd: StateDict, prefix: str, x: Tensor,
                  keep_mask: Optional[Tensor], p: float) -> Tensor:
    """Linear,GN(8),SiLU,Dropout(p),Linear,GN(8),SiLU (models/diffusion.py:198-208).

    ``keep_mask`` (0/1, same shape as the first half's output) switches the
    dropout on; ``None`` is eval mode."""
    h = _half_block(sd, prefix, 0, 1, x)
    if keep_mask is not None:
        h = h * (keep_mask / (1.0 - p))
    return _half_block(sd, prefix, 4, 5, h)


def unet_forward(sd: StateDict, x: Tensor, t_norm: Tensor, c_emb: Tensor,
                 n_hidden: int, time_dim: int,
                 keep_masks: Optional[Sequence[Tensor]] = None, p: float = 0.0,
                 taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """DiffusionUNet.forward (models/diffusion.py:210-256).

    ``keep_masks``: one 0/1 tensor per block in execution order (train mode) or
    None (eval).  ``taps`` (optional dict) receives every intermediate."""
    t_emb = F.linear(time_embedding(t_norm, time_dim),
                     sd["unet.time_proj.weight"], sd["unet.time_proj.bias"])
    c_proj = F.linear(c_emb, sd["unet.cond_proj.weight"], sd["unet.cond_proj.bias"])
    h = F.linear(x, sd["unet.input_proj.weight"], sd["unet.input_proj.bias"])
    h = h + t_emb + c_proj
    if taps is not None:
        taps["t_emb"], taps["c_proj"], taps["h0"] = t_emb, c_proj, h
    names = block_names(n_hidden)
    n_enc = n_hidden - 1
    skips: List[Tensor] = []
    for bi, name in enumerate(names):
        if bi > n_enc:                       # decoder: concat with LIFO skip
            h = torch.cat([h, skips.pop()], dim=-1)
        mask = None if keep_masks is None else keep_masks[bi]
        h = block_forward(sd, name, h, mask, p)
        if bi < n_enc:
            skips.append(h)
        if taps is not None:
            taps[name] = h
    return F.linear(h, sd["unet.output_proj.weight"], sd["unet.output_proj.bias"])


# --------------------------------------------------------------------------
# diffusion process  (models/diffusion.py:328-449)
# --------------------------------------------------------------------------
def q_sample(bufs: Dict[str, Tensor], x0: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """x_t = sqrt(abar_t) x0 + sqrt(1-abar_t) eps (models/diffusion.py:337-340)."""
    a = bufs["sqrt_alphas_cumprod"][t].view(-1, 1).to(x0.dtype)
    b = bufs["sqrt_one_minus_alphas_cumprod"][t].view(-1, 1).to(x0.dtype)
    return a * x0 + b * noise


def training_forward(sd: StateDict, bufs: Dict[str, Tensor], x0: Tensor, cond: Tensor,
                     t: Tensor, noise: Tensor, n_hidden: int, time_dim: int,
                     keep_masks: Optional[Sequence[Tensor]] = None, p: float = 0.0,
                     return_loss: bool = True) -> Tensor:
    """BiologyAwareDiffusionModel.forward with t / noise / masks injected
    (models/diffusion.py:344-380)."""
    T = bufs["betas"].shape[0]
    x_t = q_sample(bufs, x0, t, noise)
    t_norm = t.to(x0.dtype) / T            # t.float() / num_steps at fp32
    c_emb = condition_embed(sd, cond)
    pred = unet_forward(sd, x_t, t_norm, c_emb, n_hidden, time_dim, keep_masks, p)
    if return_loss:
        return F.mse_loss(pred, noise)
    return pred


def training_loss_and_grads(sd: StateDict, bufs, x0, cond, t, noise, n_hidden, time_dim,
                            keep_masks=None, p: float = 0.0):
    """Loss and dLoss/dparam for every entry of ``sd`` (autograd over the ops above;
    what loss.backward() does at utils/train.py:239)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    loss = training_forward(leaves, bufs, x0, cond, t, noise, n_hidden, time_dim, keep_masks, p)
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    out = {}
    for (k, v), g in zip(leaves.items(), grads):
        out[k] = torch.zeros_like(v) if g is None else g
    return loss.detach(), out


def p_sample(sd: StateDict, bufs, x_t: Tensor, t: int, cond: Tensor, z: Optional[Tensor],
             n_hidden: int, time_dim: int,
             keep_masks: Optional[Sequence[Tensor]] = None, p: float = 0.0) -> Tensor:
    """One reverse step; op order of models/diffusion.py:392-425 (no x0 clamp)."""
    T = bufs["betas"].shape[0]
    dt = x_t.dtype
    n = x_t.shape[0]
    t_norm = torch.full((n,), t / T).to(dt)
    c_emb = condition_embed(sd, cond)
    eps = unet_forward(sd, x_t, t_norm, c_emb, n_hidden, time_dim, keep_masks, p)
    betas = bufs["betas"].to(dt)
    abar = bufs["alphas_cumprod"].to(dt)
    alpha_t = 1.0 - betas[t]
    ab = abar[t]
    x0_pred = (x_t - torch.sqrt(1 - ab) * eps) / torch.sqrt(ab)
    if t > 0:
        abp = abar[t - 1]
        mean = (torch.sqrt(abp) * betas[t] * x0_pred / (1 - ab)
                + torch.sqrt(alpha_t) * (1 - abp) * x_t / (1 - ab))
        var = (1 - abp) / (1 - ab) * betas[t]
        return mean + torch.sqrt(var) * z
    return x0_pred


def sample(sd: StateDict, bufs, cond: Tensor, x_T: Tensor,
           noise_fn: Callable[[int], Optional[Tensor]], n_hidden: int, time_dim: int,
           mask_fn: Optional[Callable[[int], Sequence[Tensor]]] = None, p: float = 0.0,
           trace: Optional[Callable[[int, Tensor], None]] = None) -> Tensor:
    """Full reverse chain (models/diffusion.py:427-449).  ``noise_fn(t)`` supplies the
    z drawn at step t (t = T-1 .. 1; never called for t == 0)."""
    T = bufs["betas"].shape[0]
    x = x_T
    for t in reversed(range(T)):
        z = noise_fn(t) if t > 0 else None
        masks = mask_fn(t) if mask_fn is not None else None
        x = p_sample(sd, bufs, x, t, cond, z, n_hidden, time_dim, masks, p)
        if trace is not None:
            trace(t, x)
    return x


# --------------------------------------------------------------------------
# generation glue  (utils/generate.py:39-144)
# --------------------------------------------------------------------------
def create_conditions(condition_on: Sequence[str], condition_dim: int, num_samples: int,
                      scenario: Optional[dict]) -> Optional[np.ndarray]:
    """Constant condition rows for a scenario (utils/generate.py:56-90); returns None
    when no scenario is given (the reference then draws randn)."""
    if scenario is None:
        return None
    vals: List[float] = []
    for name in condition_on:
        if name == "survival_time":
            vals.append((scenario.get("survival_time", 800) - 800) / 500)
        elif name == "event_occurred":
            vals.append(scenario.get("event_occurred", 0))
        elif name == "age":
            vals.append(scenario.get("age", 15.0))
        elif name == "metastasis_at_diagnosis":
            vals.append(scenario.get("metastasis_at_diagnosis", 0))
    if len(vals) < condition_dim:
        vals = vals + [0.0] * (condition_dim - len(vals))
    else:
        vals = vals[:condition_dim]
    return np.tile(np.asarray(vals, dtype=np.float32)[None, :], (num_samples, 1))


def split_and_binarize(samples: np.ndarray, mutation_dim: int, expression_dim: int):
    """Column split + (mutations > 0.5).astype(float) (utils/generate.py:130-135)."""
    mut = samples[:, :mutation_dim]
    expr = samples[:, mutation_dim:mutation_dim + expression_dim]
    path = samples[:, mutation_dim + expression_dim:]
    return (mut > 0.5).astype(float), expr, path


# --------------------------------------------------------------------------
# training-loop arithmetic  (utils/train.py:91-126, 204-250)
# --------------------------------------------------------------------------
def mixup(data: Tensor, cond: Tensor, surv: Tensor, lam: float, perm: Tensor):
    """lam*x + (1-lam)*x[perm] on data, conditions, survival (utils/train.py:117-120)."""
    return (lam * data + (1 - lam) * data[perm],
            lam * cond + (1 - lam) * cond[perm],
            lam * surv + (1 - lam) * surv[perm])


def clip_grad_norm(grads: Sequence[Tensor], max_norm: float):
    """torch.nn.utils.clip_grad_norm_ semantics (utils/train.py:242): global L2 norm,
    coefficient max_norm/(norm+1e-6) clamped to 1.  Returns (clipped grads, norm)."""
    norms = torch.stack([torch.linalg.vector_norm(g, 2.0) for g in grads])
    total = torch.linalg.vector_norm(norms, 2.0)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], total


def adamw_step(params: Sequence[Tensor], grads: Sequence[Tensor],
               exp_avg: Sequence[Tensor], exp_avg_sq: Sequence[Tensor], step: int,
               lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
               weight_decay: float = 1e-2):
    """One torch.optim.AdamW update (utils/train.py:169-173,244), single-tensor form.
    ``step`` is the 1-based step count AFTER increment.  Updates in place."""
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        p.mul_(1 - lr * weight_decay)
        m.lerp_(g, 1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))


# --------------------------------------------------------------------------
# helpers shared by tests / bench cpu_baseline
# --------------------------------------------------------------------------
def param_shapes(mutation_dim: int, expression_dim: int, pathway_dim: int, condition_dim: int,
                 hidden_dims: Sequence[int], time_dim: int) -> "Dict[str, tuple]":
    """Ordered {state_dict key: shape} for the trainable parameters
    (constructor order of models/diffusion.py:283-295, 160-196)."""
    D = mutation_dim + expression_dim + pathway_dim
    H = list(hidden_dims)
    shapes: Dict[str, tuple] = {}
    shapes["condition_embed.mlp.0.weight"] = (COND_EMBED_WIDTH, condition_dim)
    shapes["condition_embed.mlp.0.bias"] = (COND_EMBED_WIDTH,)
    shapes["condition_embed.mlp.2.weight"] = (COND_EMBED_WIDTH, COND_EMBED_WIDTH)
    shapes["condition_embed.mlp.2.bias"] = (COND_EMBED_WIDTH,)
    shapes["unet.input_proj.weight"] = (H[0], D)
    shapes["unet.input_proj.bias"] = (H[0],)
    shapes["unet.cond_proj.weight"] = (H[0], time_dim // 2)
    shapes["unet.cond_proj.bias"] = (H[0],)
    shapes["unet.time_proj.weight"] = (H[0], time_dim)
    shapes["unet.time_proj.bias"] = (H[0],)

    def block(prefix, cin, cout):
        shapes[f"{prefix}.0.weight"] = (cout, cin)
        shapes[f"{prefix}.0.bias"] = (cout,)
        shapes[f"{prefix}.1.weight"] = (cout,)
        shapes[f"{prefix}.1.bias"] = (cout,)
        shapes[f"{prefix}.4.weight"] = (cout, cout)
        shapes[f"{prefix}.4.bias"] = (cout,)
        shapes[f"{prefix}.5.weight"] = (cout,)
        shapes[f"{prefix}.5.bias"] = (cout,)

    cin = H[0]
    for i, h in enumerate(H[1:]):
        block(f"unet.encoder.{i}", cin, h)
        cin = h
    block("unet.bottleneck", cin, cin)
    cur = H[-1]
    for j, i in enumerate(range(len(H) - 2, -1, -1)):
        block(f"unet.decoder.{j}", cur + H[i + 1], H[i])
        cur = H[i]
    shapes["unet.output_proj.weight"] = (D, cur)
    shapes["unet.output_proj.bias"] = (D,)
    return shapes


def init_state_dict(shapes: "Dict[str, tuple]", seed: int = 0) -> StateDict:
    """Deterministic nn.Linear/GroupNorm-style init from OUR OWN generator recipe
    (uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)); GroupNorm weight 1, bias 0).  Not
    bit-identical to torch.manual_seed(0) module construction -- it only needs the
    same distribution; parity tests load the SAME dict into both sides."""
    g = torch.Generator().manual_seed(seed)
    sd: StateDict = {}
    fan_in = 1
    for k, shp in shapes.items():
        parts = k.split(".")
        is_gn = parts[-2] in ("1", "5") and len(shp) == 1 and (
            "encoder" in k or "decoder" in k or "bottleneck" in k)
        if is_gn:
            sd[k] = torch.ones(shp) if parts[-1] == "weight" else torch.zeros(shp)
            continue
        if parts[-1] == "weight":
            fan_in = shp[1]
        bound = 1.0 / math.sqrt(fan_in)
        sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * bound
    return sd


def to_dtype(sd: StateDict, dtype) -> StateDict:
    return {k: v.to(dtype) for k, v in sd.items()}
