"""CPU oracle for the cVAE (models/cvae.py) -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Functional restatement over a ``state_dict`` with plain torch CPU ops, every random draw injected; pinned to the
reference's own ``BiologyConstrainedVAE`` by tests/golden/g10_cvae.npz (tests/golden/make_goldens.py g10).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]


def _mlp(sd: StateDict, prefix: str, h: Tensor, n_layers: int, training: bool, masks: Optional[Sequence[Tensor]], p: float,
         new_stats: Optional[StateDict], momentum: float = 0.1, eps: float = 1e-5) -> Tensor:
    """Linear -> BatchNorm1d -> ReLU -> Dropout groups (models/cvae.py:26-35, 76-85); module indices 4k, 4k+1."""
    for k in range(n_layers):
        lin, bn = f"{prefix}.{4 * k}", f"{prefix}.{4 * k + 1}"
        z = F.linear(h, sd[f"{lin}.weight"], sd[f"{lin}.bias"])
        if training:
            mean = z.mean(dim=0)
            var_b = z.var(dim=0, unbiased=False)
            if new_stats is not None:
                n = z.shape[0]
                new_stats[f"{bn}.running_mean"] = (1 - momentum) * sd[f"{bn}.running_mean"] + momentum * mean.detach()
                new_stats[f"{bn}.running_var"] = (1 - momentum) * sd[f"{bn}.running_var"] + momentum * z.var(dim=0, unbiased=True).detach()
                new_stats[f"{bn}.num_batches_tracked"] = sd[f"{bn}.num_batches_tracked"] + 1
                del n
        else:
            mean, var_b = sd[f"{bn}.running_mean"], sd[f"{bn}.running_var"]
        y = (z - mean) / torch.sqrt(var_b + eps) * sd[f"{bn}.weight"] + sd[f"{bn}.bias"]
        h = F.relu(y)
        if training and p > 0:
            h = h * masks[k] / (1.0 - p)
    return h


def n_hidden_layers(sd: StateDict, prefix: str) -> int:
    k = 0
    while f"{prefix}.{4 * k}.weight" in sd:
        k += 1
    return k


def encode(sd: StateDict, x: Tensor, cond: Tensor, training: bool, masks=None, p: float = 0.0, new_stats=None, pre: str = "vae.") -> Tuple[Tensor, Tensor]:
    """models/cvae.py:43-60."""
    h = torch.cat([x, cond], dim=-1)
    h = _mlp(sd, f"{pre}encoder.mlp", h, n_hidden_layers(sd, f"{pre}encoder.mlp"), training, masks, p, new_stats)
    return (F.linear(h, sd[f"{pre}encoder.fc_mu.weight"], sd[f"{pre}encoder.fc_mu.bias"]),
            F.linear(h, sd[f"{pre}encoder.fc_logvar.weight"], sd[f"{pre}encoder.fc_logvar.bias"]))


def decode(sd: StateDict, z: Tensor, cond: Tensor, training: bool, masks=None, p: float = 0.0, new_stats=None, pre: str = "vae.") -> Tensor:
    """models/cvae.py:88-100."""
    h = torch.cat([z, cond], dim=-1)
    h = _mlp(sd, f"{pre}decoder.mlp", h, n_hidden_layers(sd, f"{pre}decoder.mlp"), training, masks, p, new_stats)
    return F.linear(h, sd[f"{pre}decoder.output.weight"], sd[f"{pre}decoder.output.bias"])


def vae_forward(sd: StateDict, x: Tensor, cond: Tensor, eps: Tensor, training: bool, enc_masks=None, dec_masks=None, p: float = 0.0,
                new_stats=None, pre: str = "vae."):
    """ConditionalVAE.forward(return_parts=True) with eps injected (models/cvae.py:158-186)."""
    mu, logvar = encode(sd, x, cond, training, enc_masks, p, new_stats, pre)
    z = mu + eps * torch.exp(0.5 * logvar)
    x_recon = decode(sd, z, cond, training, dec_masks, p, new_stats, pre)
    recon = F.mse_loss(x_recon, x, reduction="sum") / x.shape[0]
    kl = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp()) / x.shape[0]
    return recon + kl, x_recon, mu, logvar, recon, kl


def constrained_forward(sd: StateDict, x: Tensor, cond: Tensor, survival: Optional[Tensor], eps: Tensor, training: bool, enc_masks=None,
                        dec_masks=None, surv_mask=None, p: float = 0.0, survival_weight: float = 0.3, new_stats=None):
    """BiologyConstrainedVAE.forward (models/cvae.py:304-338); the two constraint stubs contribute 0.0."""
    loss, x_recon, mu, logvar, recon, kl = vae_forward(sd, x, cond, eps, training, enc_masks, dec_masks, p, new_stats)
    if survival is not None:
        h = F.relu(F.linear(mu, sd["survival_predictor.0.weight"], sd["survival_predictor.0.bias"]))
        if training:
            h = h * surv_mask / (1.0 - 0.2)                     # nn.Dropout(0.2), models/cvae.py:252
        pred = F.linear(h, sd["survival_predictor.3.weight"], sd["survival_predictor.3.bias"]).squeeze()
        loss = loss + survival_weight * F.mse_loss(pred, survival)
    return loss, x_recon, mu, logvar, recon, kl


def loss_and_grads(sd: StateDict, *args, **kwargs):
    """Loss and d loss / d parameter for every floating-point non-buffer entry (what loss.backward() gives)."""
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k]
    leaves = dict(sd)
    for k in names:
        leaves[k] = sd[k].detach().clone().requires_grad_(True)
    out = constrained_forward(leaves, *args, **kwargs)
    grads = torch.autograd.grad(out[0], [leaves[k] for k in names], allow_unused=True)
    return out, {k: (torch.zeros_like(leaves[k]) if g is None else g) for k, g in zip(names, grads)}
