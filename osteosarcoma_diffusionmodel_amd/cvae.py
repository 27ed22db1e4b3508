"""MI355X mirror of models/cvae.py: ``Encoder``, ``Decoder``, ``ConditionalVAE`` and
``BiologyConstrainedVAE`` with the reference's constructor arguments, attribute names and ``state_dict`` keys
(``vae.encoder.mlp.{0,1,4,5,..}``, ``vae.encoder.fc_mu``, ``vae.decoder.output``, ``survival_predictor.{0,3}`` ...).

The ``nn.Linear`` / ``nn.BatchNorm1d`` sub-modules only own the parameters and buffers; every forward and backward runs
through the HIP layer ops of libosdiff.so (``nn_ops``): concat-free Linear on the MFMA GEMM kernels,
BatchNorm1d + ReLU + Dropout with batch statistics as double column sums, reparameterisation and the VAE loss.
Keyword-only additions (``eps=``, ``dropout_masks=``, ``seed=``, ``z=``) inject the random draws for parity tests;
without them dropout and eps come from the library's Philox stream seeded from torch's default generator.

The two constraint losses of ``BiologyConstrainedVAE`` return 0.0 as in the reference (models/cvae.py:262-302);
``set_constraints`` switches on this package's definitions (``constraints.py``) for the reconstruction.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import nn_ops as F
from .constraints import mutation_expression_correlation_loss, pathway_coherence_loss
from .diffusion import _draw_seed


def _mlp(in_dim: int, dims: Sequence[int], dropout: float) -> nn.Sequential:
    layers: List[nn.Module] = []
    for h in dims:
        layers.extend([nn.Linear(in_dim, h), nn.BatchNorm1d(h), nn.ReLU(), nn.Dropout(dropout)])
        in_dim = h
    return nn.Sequential(*layers)


def _run_mlp(mlp: nn.Sequential, x1, x2, training: bool, masks, seed: int, tag0: int):
    """Linear -> BatchNorm1d -> ReLU -> Dropout groups of four; the first Linear takes the concat-free pair."""
    h = x1
    for i in range(0, len(mlp), 4):
        lin, bn, drop = mlp[i], mlp[i + 1], mlp[i + 3]
        z = F.linear(h, x2 if i == 0 else None, lin.weight, lin.bias)
        k = i // 4
        h = F.bn_relu_dropout(z, bn, training, drop.p, None if masks is None else masks[k], seed, tag0 + k)
        if training and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
    return h


class Encoder(nn.Module):
    """models/cvae.py:12-60."""

    def __init__(self, input_dim: int, condition_dim: int, hidden_dims: list, latent_dim: int, dropout: float = 0.1):
        super().__init__()
        self.mlp = _mlp(input_dim + condition_dim, hidden_dims, dropout)
        last = hidden_dims[-1] if hidden_dims else input_dim + condition_dim
        self.fc_mu = nn.Linear(last, latent_dim)
        self.fc_logvar = nn.Linear(last, latent_dim)

    def forward(self, x, conditions, *, dropout_masks=None, seed: int = 0):
        if len(self.mlp) == 0:
            raise ValueError("hidden_dims must not be empty")
        h = _run_mlp(self.mlp, x, conditions, self.training, dropout_masks, seed, 0x10)
        return F.linear(h, None, self.fc_mu.weight, self.fc_mu.bias), F.linear(h, None, self.fc_logvar.weight, self.fc_logvar.bias)


class Decoder(nn.Module):
    """models/cvae.py:63-100."""

    def __init__(self, latent_dim: int, condition_dim: int, hidden_dims: list, output_dim: int, dropout: float = 0.1):
        super().__init__()
        rev = list(reversed(hidden_dims))
        self.mlp = _mlp(latent_dim + condition_dim, rev, dropout)
        self.output = nn.Linear(rev[-1] if rev else latent_dim + condition_dim, output_dim)

    def forward(self, z, conditions, *, dropout_masks=None, seed: int = 0):
        if len(self.mlp) == 0:
            raise ValueError("hidden_dims must not be empty")
        h = _run_mlp(self.mlp, z, conditions, self.training, dropout_masks, seed, 0x20)
        return F.linear(h, None, self.output.weight, self.output.bias)


class ConditionalVAE(nn.Module):
    """models/cvae.py:103-219."""

    def __init__(self, mutation_dim: int, expression_dim: int, pathway_dim: int, condition_dim: int, config: dict):
        super().__init__()
        self.mutation_dim, self.expression_dim, self.pathway_dim = mutation_dim, expression_dim, pathway_dim
        self.condition_dim = condition_dim
        self.data_dim = mutation_dim + expression_dim + pathway_dim
        m = config["model"]
        self.latent_dim = m["latent_dim"]
        self.encoder = Encoder(self.data_dim, condition_dim, m["hidden_dims"], self.latent_dim, m["gnn"]["dropout"])
        self.decoder = Decoder(self.latent_dim, condition_dim, m["hidden_dims"], self.data_dim, m["gnn"]["dropout"])

    def _check(self, x, conditions):
        if x.device.type != "cuda":
            raise RuntimeError("ConditionalVAE (osteosarcoma_diffusionmodel_amd) runs on MI355X only: move the model and its "
                               "inputs to a ROCm device; there is no CPU fallback")
        if x.dim() != 2 or conditions.dim() != 2 or x.shape[0] != conditions.shape[0]:
            raise RuntimeError(f"expected [N, D] data and [N, C] conditions, got {tuple(x.shape)} and {tuple(conditions.shape)}")

    def reparameterize(self, mu, logvar, *, eps=None, seed: Optional[int] = None):
        return F.reparameterize(mu, logvar, eps, _draw_seed() if (seed is None and eps is None) else (seed or 0))

    def forward(self, x, conditions, return_parts=False, *, eps=None, dropout_masks=None, seed: Optional[int] = None):
        """dropout_masks: (encoder masks, decoder masks), each a list of [N, h] 0/1 keep-masks, or None."""
        self._check(x, conditions)
        seed = _draw_seed() if seed is None else seed
        em, dm = dropout_masks if dropout_masks is not None else (None, None)
        mu, logvar = self.encoder(x, conditions, dropout_masks=em, seed=seed)
        z = self.reparameterize(mu, logvar, eps=eps, seed=seed)
        x_recon = self.decoder(z, conditions, dropout_masks=dm, seed=seed)
        loss, recon_loss, kl_loss = F.vae_loss(x_recon, x, mu, logvar)
        if return_parts:
            return loss, x_recon, mu, logvar, recon_loss, kl_loss
        return loss

    @torch.no_grad()
    def sample(self, conditions, num_samples: int = 1, *, z=None):
        device = next(self.parameters()).device
        if z is None:
            z = torch.randn(num_samples, self.latent_dim, device=device)
        return self.decoder(z, conditions)

    @torch.no_grad()
    def encode(self, x, conditions):
        mu, _ = self.encoder(x, conditions)
        return mu

    @torch.no_grad()
    def decode(self, z, conditions):
        return self.decoder(z, conditions)


class BiologyConstrainedVAE(nn.Module):
    """models/cvae.py:222-345."""

    def __init__(self, mutation_dim: int, expression_dim: int, pathway_dim: int, condition_dim: int, config: dict):
        super().__init__()
        self.vae = ConditionalVAE(mutation_dim, expression_dim, pathway_dim, condition_dim, config)
        self.mutation_dim, self.expression_dim, self.pathway_dim = mutation_dim, expression_dim, pathway_dim
        self.condition_dim = condition_dim          # not in the reference; SyntheticPatientGenerator reads it (utils/generate.py:37)
        self.survival_predictor = nn.Sequential(nn.Linear(config["model"]["latent_dim"], 128), nn.ReLU(), nn.Dropout(0.2), nn.Linear(128, 1))
        c = config["model"]["constraints"]
        self.pathway_coherence_weight = c["pathway_coherence_weight"]
        self.mutation_expr_weight = c["mutation_expression_weight"]
        self.survival_weight = c["survival_prediction_weight"]
        self._pathways = None
        self._me_cols = None

    def set_constraints(self, pathways=None, mutation_columns=None, target_columns=None):
        """Switch the two constraint terms from the reference's 0.0 stubs to this package's definitions."""
        self._pathways = [list(map(int, p)) for p in pathways] if pathways else None
        self._me_cols = (list(map(int, mutation_columns)), list(map(int, target_columns))) if mutation_columns else None

    def pathway_coherence_loss(self, x_recon, pathway_gene_matrix=None):
        if self._pathways is None:
            return 0.0                                   # models/cvae.py:279-281
        return pathway_coherence_loss(x_recon, self._pathways)

    def mutation_expression_correlation_loss(self, x_recon, x_true):
        if self._me_cols is None:
            return 0.0                                   # models/cvae.py:298-300
        return mutation_expression_correlation_loss(x_recon, x_true, *self._me_cols)

    def forward(self, x, conditions, survival_time=None, *, eps=None, dropout_masks=None, survival_mask=None, seed: Optional[int] = None):
        seed = _draw_seed() if seed is None else seed
        vae_loss, x_recon, mu, logvar, recon_loss, kl_loss = self.vae(x, conditions, return_parts=True, eps=eps, dropout_masks=dropout_masks,
                                                                      seed=seed)
        pathway_loss = self.pathway_coherence_loss(x_recon, None)
        mut_expr_loss = self.mutation_expression_correlation_loss(x_recon, x)
        if survival_time is not None:
            sp = self.survival_predictor
            h = F.linear(mu, None, sp[0].weight, sp[0].bias)
            h = F.bn_relu_dropout(h, None, self.training, sp[2].p, survival_mask, seed, 0x30)
            pred = F.linear(h, None, sp[3].weight, sp[3].bias).squeeze()
            survival_loss = F.mse_mean(pred, survival_time.to(pred.dtype))
        else:
            survival_loss = 0.0
        return (vae_loss + self.pathway_coherence_weight * pathway_loss + self.mutation_expr_weight * mut_expr_loss
                + self.survival_weight * survival_loss)

    @torch.no_grad()
    def sample(self, conditions, num_samples: int = 1, *, z=None):
        return self.vae.sample(conditions, num_samples, z=z)
