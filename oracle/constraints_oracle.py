"""CPU oracle for the biological constraint losses -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

The reference declares these losses as stubs that return 0.0 (models/cvae.py:262-302: ``pathway_coherence_loss``,
``mutation_expression_correlation_loss``) and the diffusion model (models/diffusion.py) has no such terms, so there is
nothing in the reference to pin the non-zero definitions to: **parity unpinned** for them.  What is pinned is the
default: with no constraints configured the training loss is the reference's eps-MSE (tests/golden g3g4/g6).

The definitions restated here in plain torch ops (gradients by autograd) are the ones include/osdiff.h documents:

  pathway coherence    L_pc = mean_P (1 - c_P), c_P = mean of the strict upper triangle of the Pearson matrix of the
                       member columns of pathway P over the batch rows -- the quantity the reference's validator
                       reports (utils/validation.py:156-161); pathways with < 2 members are skipped
  mutation-expression  L_me = mean_{i in A, j in B} (corr_recon(i, j) - corr_true(i, j))^2 -- the docstring's "MSE on
                       correlation matrices" (models/cvae.py:296-297)

Constant columns are given a standardised value of 0 (correlation 0 with everything) instead of NaN.
"""
from __future__ import annotations

from typing import List, Sequence

import torch


def _standardise(x: torch.Tensor) -> torch.Tensor:
    """(x - mean) / std over the rows, ddof = 1 (as pandas .corr()); constant columns -> 0."""
    mu = x.mean(dim=0, keepdim=True)
    var = ((x - mu) ** 2).sum(dim=0, keepdim=True) / (x.shape[0] - 1)
    alive = var > 1e-12 * (x ** 2).mean(dim=0, keepdim=True)
    sd = torch.where(alive, var, torch.ones_like(var)).sqrt()
    return torch.where(alive, (x - mu) / sd, torch.zeros_like(x))


def correlation_block(x: torch.Tensor, cols_a: Sequence[int], cols_b: Sequence[int]) -> torch.Tensor:
    z = _standardise(x)
    return z[:, list(cols_a)].T @ z[:, list(cols_b)] / (x.shape[0] - 1)


def pathway_coherence_loss(x: torch.Tensor, pathways: List[Sequence[int]]) -> torch.Tensor:
    """x [rows, D]; pathways: list of member-column lists."""
    terms = []
    for members in pathways:
        g = len(members)
        if g < 2:
            continue
        c = correlation_block(x, members, members)
        iu = torch.triu_indices(g, g, offset=1)
        terms.append(1.0 - c[iu[0], iu[1]].mean())
    if not terms:
        return x.sum() * 0.0
    return torch.stack(terms).mean()


def mutation_expression_correlation_loss(x_recon: torch.Tensor, x_true: torch.Tensor, cols_a: Sequence[int],
                                         cols_b: Sequence[int]) -> torch.Tensor:
    cr = correlation_block(x_recon, cols_a, cols_b)
    ct = correlation_block(x_true, cols_a, cols_b).detach()
    return ((cr - ct) ** 2).mean()


def x0_hat(x_t: torch.Tensor, eps_hat: torch.Tensor, t: torch.Tensor, sqrt_ac: torch.Tensor, sqrt_1m: torch.Tensor) -> torch.Tensor:
    """models/diffusion.py:405 with per-row t."""
    return (x_t - sqrt_1m[t][:, None] * eps_hat) / sqrt_ac[t][:, None]
