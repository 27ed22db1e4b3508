"""Shared helpers for the GPU parity tests."""
import ctypes as C

import numpy as np
import torch

from osteosarcoma_diffusionmodel_amd import _lib as L
from osteosarcoma_diffusionmodel_amd.diffusion import BiologyAwareDiffusionModel

SM = dict(mutation_dim=8, expression_dim=24, pathway_dim=8, condition_dim=3)
SM_H = [32, 64, 32]
FULL = dict(mutation_dim=50, expression_dim=1900, pathway_dim=50, condition_dim=3)
FULL_H = [256, 512, 256]


def config(hidden, T=1000, schedule="cosine", p=0.2):
    return {"model": {"latent_dim": 128, "hidden_dims": list(hidden), "gnn": {"dropout": p},
                      "diffusion": {"num_steps": T, "beta_schedule": schedule},
                      "condition_on": ["survival_time", "event_occurred", "metastasis_at_diagnosis"],
                      "architecture": "diffusion"}}


def load_golden(golden_dir, name):
    return dict(np.load(golden_dir / f"{name}.npz"))


def golden_small_sd(golden_dir):
    g = load_golden(golden_dir, "g3g4_small_model")
    return {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}


def small_model(golden_dir, T=1000, p=0.2, device="cuda"):
    m = BiologyAwareDiffusionModel(config=config(SM_H, T=T, p=p), **SM)
    sd = golden_small_sd(golden_dir)
    params = {k: v for k, v in sd.items() if k.startswith(("condition_embed", "unet"))}
    m.load_state_dict(params, strict=False)      # schedule buffers depend on T: keep the model's own
    return m.to(device).eval()


def rel_err(a, b):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _np64(v):
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu()
    return np.asarray(v, dtype=np.float64)


def assert_close(a, b, rtol, atol=0.0, what=""):
    a = _np64(a)
    b = _np64(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    tol = atol + rtol * np.abs(b).max()
    err = np.abs(a - b).max()
    assert np.isfinite(a).all(), f"{what}: non-finite output"
    assert err <= tol, f"{what}: max|d|={err:.3e} > tol={tol:.3e} (max|ref|={np.abs(b).max():.3e})"


class RawHandle:
    """A bare osd_handle for building-block tests (no weights needed)."""

    def __init__(self):
        cfg = L.OsdConfig()
        cfg.mutation_dim, cfg.expression_dim, cfg.pathway_dim, cfg.condition_dim = 8, 24, 8, 3
        cfg.time_dim, cfg.n_hidden = 128, 3
        for i, v in enumerate(SM_H):
            cfg.hidden_dims[i] = v
        cfg.num_steps, cfg.dropout_p, cfg.device = 10, 0.0, torch.cuda.current_device()
        self.cfg = cfg
        self.h = C.c_void_p()
        L.check(L.lib().osd_create(C.byref(cfg), C.byref(self.h)))
        L.check(L.lib().osd_set_stream(self.h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def close(self):
        if self.h:
            L.lib().osd_destroy(self.h)
            self.h = C.c_void_p()


# ---- numpy restatement of the library's Philox4x32-10 addressing (csrc/rng.h) -------------
TAG_POSTERIOR, TAG_QNOISE, TAG_TSTEP, TAG_DROPOUT = 0x50535400, 0x514E5300, 0x54535400, 0x44524F00


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (Salmon et al. 2011); all inputs uint32 arrays / scalars."""
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint32) for v in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = int(k0), int(k1)
    for _ in range(10):
        p0 = M0 * c0.astype(np.uint64)
        p1 = M1 * c2.astype(np.uint64)
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint32(k0), lo1, hi0 ^ c3 ^ np.uint32(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def philox_block(seed, rows, cols, step, tag, row_offset=0):
    """uint32 [rows][cols] as the kernels draw them: element (r, c) is word c%4 of the block at
    counter (row_offset + r, c//4, step, tag), key = (seed_lo, seed_hi)."""
    c4 = (cols + 3) // 4
    r = (np.arange(rows, dtype=np.uint64)[:, None] + np.uint64(row_offset)).astype(np.uint32)
    q = np.arange(c4, dtype=np.uint32)[None, :]
    w = philox4x32_10(r, q, np.uint32(step), np.uint32(tag), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    out = np.stack(w, axis=-1).reshape(rows, 4 * c4)
    return out[:, :cols]


def philox_keep_mask(seed, rows, cols, block, p, row_offset=0, step=0):
    u = (philox_block(seed, rows, cols, step, TAG_DROPOUT + block, row_offset) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return (u >= np.float32(p)).astype(np.float32)


def philox_normals(seed, rows, cols, step, tag, row_offset=0):
    """Box-Muller as csrc/rng.h:normal4 (float64 here; the device uses fast fp32 log/sin/cos)."""
    w = philox_block(seed, rows, (cols + 3) // 4 * 4, step, tag, row_offset).astype(np.float64).reshape(rows, -1, 4)
    u1a = (w[..., 0] + 0.5) * 2.0 ** -32
    u1b = (w[..., 2] + 0.5) * 2.0 ** -32
    a0 = np.floor(w[..., 1] / 256.0) * 2.0 ** -24
    a1 = np.floor(w[..., 3] / 256.0) * 2.0 ** -24
    r0, r1 = np.sqrt(-2 * np.log(u1a)), np.sqrt(-2 * np.log(u1b))
    out = np.stack([r0 * np.cos(2 * np.pi * a0), r0 * np.sin(2 * np.pi * a0), r1 * np.cos(2 * np.pi * a1), r1 * np.sin(2 * np.pi * a1)], axis=-1)
    return out.reshape(rows, -1)[:, :cols]
