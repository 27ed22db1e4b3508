"""Differentiable layer ops over libosdiff.so (``osd_nn_*``, include/osdiff.h) used by the cVAE mirror.

Each op is a ``torch.autograd.Function`` whose forward and backward are HIP kernels; torch only owns the tensors and the
graph.  Device tensors only -- there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L


def _ctx(t: torch.Tensor):
    if t.device.type != "cuda":
        raise RuntimeError("osteosarcoma_diffusionmodel_amd layer ops run on a ROCm device only; there is no CPU fallback")
    dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream), dev


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float32).contiguous()


class _Linear(torch.autograd.Function):
    """y = cat([x1, x2], -1) @ w.T + b   (x2 may be None; x2 receives no gradient)."""

    @staticmethod
    def forward(ctx, x1, x2, w, b):
        x1, w, b = _f32(x1), _f32(w), _f32(b)
        x2 = _f32(x2) if x2 is not None else None
        n, k1 = x1.shape
        k2 = x2.shape[1] if x2 is not None else 0
        if w.shape[1] != k1 + k2:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({n}x{k1 + k2} and {w.shape[1]}x{w.shape[0]})")
        y = torch.empty(n, w.shape[0], device=x1.device, dtype=torch.float32)
        stream, dev = _ctx(x1)
        L.check(L.lib().osd_nn_linear(stream, dev, L.ptr(x1), k1, L.ptr(x2), k2, L.ptr(w), L.ptr(b), n, w.shape[0], L.ptr(y)))
        ctx.save_for_backward(x1, x2, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        x1, x2, w = ctx.saved_tensors
        gy = _f32(gy)
        n, k1 = x1.shape
        k2 = x2.shape[1] if x2 is not None else 0
        dx1 = torch.empty_like(x1) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty(w.shape[0], device=w.device, dtype=torch.float32)
        stream, dev = _ctx(x1)
        L.check(L.lib().osd_nn_linear_bwd(stream, dev, L.ptr(x1), k1, L.ptr(x2), k2, L.ptr(w), L.ptr(gy), n, w.shape[0], L.ptr(dx1),
                                          L.ptr(dw), L.ptr(db)))
        return dx1, None, dw, db


class _BnReluDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, momentum, eps, training, p, mask, seed, tag):
        z = _f32(z)
        n, c = z.shape
        use_bn = gamma is not None
        y = torch.empty_like(z)
        save_mean = torch.empty(c, device=z.device, dtype=torch.float32) if use_bn else None
        save_invstd = torch.empty(c, device=z.device, dtype=torch.float32) if use_bn else None
        mask = _f32(mask) if mask is not None else None
        stream, dev = _ctx(z)
        L.check(L.lib().osd_nn_bn_relu_dropout(stream, dev, L.ptr(z), n, c, L.ptr(gamma), L.ptr(beta), L.ptr(running_mean), L.ptr(running_var),
                                               float(momentum), float(eps), int(training), int(use_bn), float(p), L.ptr(mask), int(seed),
                                               int(tag), L.ptr(y), L.ptr(save_mean), L.ptr(save_invstd)))
        ctx.save_for_backward(z, gamma, beta, save_mean, save_invstd, mask)
        ctx.cfg = (int(training), int(use_bn), float(p), int(seed), int(tag))
        return y

    @staticmethod
    def backward(ctx, gy):
        z, gamma, beta, save_mean, save_invstd, mask = ctx.saved_tensors
        training, use_bn, p, seed, tag = ctx.cfg
        gy = _f32(gy)
        n, c = z.shape
        dz = torch.empty_like(z)
        dgamma = torch.empty_like(gamma) if use_bn else None
        dbeta = torch.empty_like(beta) if use_bn else None
        stream, dev = _ctx(z)
        L.check(L.lib().osd_nn_bn_relu_dropout_bwd(stream, dev, L.ptr(gy), L.ptr(z), n, c, L.ptr(gamma), L.ptr(beta), L.ptr(save_mean),
                                                   L.ptr(save_invstd), training, use_bn, p, L.ptr(mask), seed, tag, L.ptr(dz), L.ptr(dgamma),
                                                   L.ptr(dbeta)))
        return dz, dgamma, dbeta, None, None, None, None, None, None, None, None, None


class _Reparameterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar, eps, seed):
        mu, logvar = _f32(mu), _f32(logvar)
        eps = _f32(eps) if eps is not None else None
        z = torch.empty_like(mu)
        stream, dev = _ctx(mu)
        L.check(L.lib().osd_nn_reparameterize(stream, dev, L.ptr(mu), L.ptr(logvar), L.ptr(eps), int(seed), mu.shape[0], mu.shape[1], L.ptr(z), None))
        ctx.save_for_backward(mu, z)
        return z

    @staticmethod
    def backward(ctx, gz):
        mu, z = ctx.saved_tensors
        gz = _f32(gz)
        dlv = torch.empty_like(mu)
        stream, dev = _ctx(mu)
        L.check(L.lib().osd_nn_reparameterize_bwd(stream, dev, L.ptr(gz), L.ptr(mu), L.ptr(z), mu.numel(), L.ptr(dlv)))
        return gz, dlv, None, None


class _VaeLoss(torch.autograd.Function):
    """(loss, recon, kl) of models/cvae.py:178-181; the gradients are produced by the same kernel pass."""

    @staticmethod
    def forward(ctx, x_recon, x, mu, logvar):
        x_recon, x, mu, logvar = _f32(x_recon), _f32(x), _f32(mu), _f32(logvar)
        n, d = x.shape
        parts = torch.empty(3, device=x.device, dtype=torch.float32)
        need = any(ctx.needs_input_grad)
        d_recon = torch.empty_like(x_recon) if need else None
        d_mu = torch.empty_like(mu) if need else None
        d_lv = torch.empty_like(mu) if need else None
        stream, dev = _ctx(x)
        L.check(L.lib().osd_nn_vae_loss(stream, dev, L.ptr(x_recon), L.ptr(x), L.ptr(mu), L.ptr(logvar), n, d, mu.shape[1], L.ptr(parts),
                                        L.ptr(d_recon), L.ptr(d_mu), L.ptr(d_lv)))
        ctx.grads = (d_recon, d_mu, d_lv)
        return parts[0], parts[1], parts[2]

    @staticmethod
    def backward(ctx, g_loss, g_recon, g_kl):
        d_recon, d_mu, d_lv = ctx.grads
        # loss = recon + kl: the upstream scalars of the parts add to the one of the total
        gr, gk = g_loss + g_recon, g_loss + g_kl
        return d_recon * gr, None, d_mu * gk, d_lv * gk


class _MseMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32(a), _f32(b)
        if a.shape != b.shape:
            raise RuntimeError(f"mse: shapes {tuple(a.shape)} and {tuple(b.shape)} differ")
        loss = torch.empty(1, device=a.device, dtype=torch.float32)
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        stream, dev = _ctx(a)
        L.check(L.lib().osd_nn_mse(stream, dev, L.ptr(a), L.ptr(b), a.numel(), L.ptr(loss), L.ptr(da)))
        ctx.da = da
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return (ctx.da * g if ctx.da is not None else None), None


def linear(x1, x2, weight, bias):
    return _Linear.apply(x1, x2, weight, bias)


def bn_relu_dropout(z, bn: Optional[torch.nn.BatchNorm1d], training: bool, p: float, mask=None, seed: int = 0, tag: int = 0):
    if bn is None:
        return _BnReluDropout.apply(z, None, None, None, None, 0.0, 0.0, training, p, mask, seed, tag)
    return _BnReluDropout.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, training, p, mask, seed, tag)


def reparameterize(mu, logvar, eps=None, seed: int = 0):
    return _Reparameterize.apply(mu, logvar, eps, seed)


def vae_loss(x_recon, x, mu, logvar):
    return _VaeLoss.apply(x_recon, x, mu, logvar)


def mse_mean(a, b):
    return _MseMean.apply(a, b)
