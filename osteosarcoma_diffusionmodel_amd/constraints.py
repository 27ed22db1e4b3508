"""Biological constraint losses on MI355X (north_star).

The reference names them ``pathway_coherence_loss`` and ``mutation_expression_correlation_loss``
(models/cvae.py:262-302) but implements both as stubs that return 0.0; the definitions used here are documented in
include/osdiff.h.  Two ways to use them:

* ``BiologyAwareDiffusionModel.set_constraints(...)`` -- the terms are added inside ``osd_train_loss_fwd_bwd``
  (evaluated on x0_hat of the batch), off by default so the default loss is the reference's eps-MSE;
* the functions below -- differentiable ops on any ``[rows, D]`` device tensor (``osd_loss_*``).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L


def csr_from_pathways(pathways: Sequence[Sequence[int]]) -> Tuple[np.ndarray, np.ndarray]:
    """Member-column lists -> (offsets int32[P+1], members int32[nnz])."""
    off = np.zeros(len(pathways) + 1, dtype=np.int32)
    for i, m in enumerate(pathways):
        off[i + 1] = off[i] + len(m)
    mem = np.fromiter((int(c) for m in pathways for c in m), dtype=np.int32, count=int(off[-1]))
    return off, mem


def pathways_from_matrix(pathway_gene_matrix, column_offset: int = 0, min_genes: int = 2) -> List[List[int]]:
    """genes x pathways 0/1 matrix (DataFrame or array, rows in feature order) -> member-column lists;
    ``column_offset`` = position of gene 0 in the feature vector (mutation_dim for the expression block)."""
    m = pathway_gene_matrix.values if hasattr(pathway_gene_matrix, "values") else np.asarray(pathway_gene_matrix)
    out = []
    for p in range(m.shape[1]):
        genes = np.nonzero(m[:, p] == 1)[0]
        if len(genes) >= min_genes:
            out.append([int(g) + column_offset for g in genes])
    return out


def _i32(a):
    arr = np.ascontiguousarray(a, dtype=np.int32)
    return arr, arr.ctypes.data_as(C.POINTER(C.c_int32))


def _check(x: torch.Tensor, name: str) -> torch.Tensor:
    if x.device.type != "cuda":
        raise RuntimeError(f"{name} must be on a ROCm device; the constraint kernels have no CPU fallback")
    if x.dim() != 2:
        raise ValueError(f"{name}: expected a [rows, D] tensor")
    return x.to(torch.float32).contiguous()


class _PathwayCoherence(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, off, mem):
        x = _check(x, "x")
        dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
        loss = torch.zeros(1, device=x.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        dx = torch.zeros_like(x) if need else None
        offa, offp = _i32(off)
        mema, memp = _i32(mem)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        L.check(L.lib().osd_loss_pathway_coherence(stream, dev, L.ptr(x), x.shape[0], x.shape[1], x.shape[1], offp, memp, len(offa) - 1,
                                                   1.0, L.ptr(loss), L.ptr(dx)))
        ctx.dx = dx
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return (ctx.dx * g if ctx.dx is not None else None), None, None


class _MutExprCorrelation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_recon, x_true, ca, cb):
        x_recon, x_true = _check(x_recon, "x_recon"), _check(x_true, "x_true")
        if x_recon.shape != x_true.shape:
            raise ValueError("x_recon and x_true must have the same shape")
        dev = x_recon.device.index if x_recon.device.index is not None else torch.cuda.current_device()
        loss = torch.zeros(1, device=x_recon.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        dx = torch.zeros_like(x_recon) if need else None
        caa, cap = _i32(ca)
        cba, cbp = _i32(cb)
        stream = C.c_void_p(torch.cuda.current_stream(x_recon.device).cuda_stream)
        L.check(L.lib().osd_loss_mutation_expression(stream, dev, L.ptr(x_recon), L.ptr(x_true), x_recon.shape[0], x_recon.shape[1],
                                                     x_recon.shape[1], cap, len(caa), cbp, len(cba), 1.0, L.ptr(loss), L.ptr(dx)))
        ctx.dx = dx
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return (ctx.dx * g if ctx.dx is not None else None), None, None, None


def pathway_coherence_loss(x: torch.Tensor, pathways: Sequence[Sequence[int]]) -> torch.Tensor:
    """L_pc = mean over pathways of (1 - mean off-diagonal Pearson correlation of the member columns of x)."""
    off, mem = csr_from_pathways(pathways)
    return _PathwayCoherence.apply(x, off, mem)


def mutation_expression_correlation_loss(x_recon: torch.Tensor, x_true: torch.Tensor, cols_a: Sequence[int],
                                         cols_b: Sequence[int]) -> torch.Tensor:
    """L_me = mean over (i in cols_a, j in cols_b) of (corr_recon(i, j) - corr_true(i, j))^2; x_true gets no gradient."""
    return _MutExprCorrelation.apply(x_recon, x_true, list(cols_a), list(cols_b))
