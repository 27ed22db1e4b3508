"""BiologyAwareDiffusionModel on MI355X -- host-side mirror of the reference's
models/diffusion.py:259-449.

Same constructor, attributes, ``state_dict`` keys and methods (``forward``,
``q_sample``, ``p_sample``, ``sample``) as the reference class, so
``utils/train.py``-style trainers and ``utils/generate.py``-style generators
work unchanged; every method runs hand-written HIP kernels through the C ABI
of ``libosdiff.so`` (include/osdiff.h).  There is no CPU path: tensors must
live on a ROCm device.

Build-only additions are keyword-only and default-off (``noise=``, ``x_T=``,
``seed=``, ``t=``, ``dropout_masks=``): they inject the random draws for parity
tests.  Without them randomness comes from the library's Philox stream, seeded
from torch's default generator (so ``torch.manual_seed`` makes runs repeatable).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L

COND_EMBED_WIDTH = 64  # literal of models/diffusion.py:285


class PathwayGraphEncoder(nn.Module):
    """Name kept importable for API parity (models/diffusion.py:14-88).  The reference
    never instantiates it and it is not part of the hot path; it needs torch_geometric."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("PathwayGraphEncoder is dead code in the reference and is not provided")


class ConditionalEmbedding(nn.Module):
    """Parameter container for Linear -> SiLU -> Linear (models/diffusion.py:91-114)."""

    def __init__(self, num_continuous: int, embedding_dim: int):
        super().__init__()
        self.num_continuous = num_continuous
        self.embedding_dim = embedding_dim
        self.mlp = nn.Sequential(nn.Linear(num_continuous, embedding_dim), nn.SiLU(),
                                 nn.Linear(embedding_dim, embedding_dim))


class TimeEmbedding(nn.Module):
    """Sinusoidal embedding of t in [0,1) (models/diffusion.py:117-139); host-side table builder."""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def forward(self, t):
        half = self.dim // 2
        step = np.log(10000) / (half - 1)
        freq = torch.exp(torch.arange(half, device=t.device) * -step)
        arg = t[:, None] * freq[None, :]
        return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)


def _block(cin: int, cout: int, p: float) -> nn.Sequential:
    # indices 0,1,4,5 carry the parameters, as in models/diffusion.py:200-208
    return nn.Sequential(nn.Linear(cin, cout), nn.GroupNorm(8, cout), nn.SiLU(), nn.Dropout(p),
                         nn.Linear(cout, cout), nn.GroupNorm(8, cout), nn.SiLU())


class DiffusionUNet(nn.Module):
    """Parameter container with the reference's module names (models/diffusion.py:142-196).
    Construction order matches the reference so a given torch seed yields the same init."""

    def __init__(self, data_dim, time_dim=128, condition_dim=64, hidden_dims=(256, 512, 256), dropout=0.1):
        super().__init__()
        hidden_dims = list(hidden_dims)
        self.data_dim = data_dim
        self.time_embed = TimeEmbedding(time_dim)
        self.input_proj = nn.Linear(data_dim, hidden_dims[0])
        self.cond_proj = nn.Linear(condition_dim, hidden_dims[0])
        self.time_proj = nn.Linear(time_dim, hidden_dims[0])
        self.encoder = nn.ModuleList()
        cin = hidden_dims[0]
        for h in hidden_dims[1:]:
            self.encoder.append(_block(cin, h, dropout))
            cin = h
        self.bottleneck = _block(cin, cin, dropout)
        self.decoder = nn.ModuleList()
        cur = hidden_dims[-1]
        for i in range(len(hidden_dims) - 2, -1, -1):
            self.decoder.append(_block(cur + hidden_dims[i + 1], hidden_dims[i], dropout))
            cur = hidden_dims[i]
        self.output_proj = nn.Linear(cur, data_dim)


def _draw_seed() -> int:
    """63-bit seed from torch's default CPU generator (follows torch.manual_seed)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


class _Engine:
    """Owns the osd_handle for one device and keeps its borrowed pointers current."""

    def __init__(self, model: "BiologyAwareDiffusionModel", device: torch.device):
        lib = L.lib()
        cfg = L.OsdConfig()
        cfg.mutation_dim, cfg.expression_dim = model.mutation_dim, model.expression_dim
        cfg.pathway_dim, cfg.condition_dim = model.pathway_dim, model.condition_dim
        cfg.time_dim = model._time_dim
        hd = model._hidden_dims
        if len(hd) > L.OSD_MAX_HIDDEN:
            raise ValueError(f"at most {L.OSD_MAX_HIDDEN} hidden dims are supported")
        cfg.n_hidden = len(hd)
        for i, v in enumerate(hd):
            cfg.hidden_dims[i] = int(v)
        cfg.num_steps = model.num_steps
        cfg.dropout_p = float(model._dropout_p)
        cfg.device = device.index if device.index is not None else torch.cuda.current_device()
        self.cfg = cfg
        self.device = device
        self.handle = C.c_void_p()
        L.check(lib.osd_create(C.byref(cfg), C.byref(self.handle)))
        self.n_params = lib.osd_num_params(C.byref(cfg))
        # schedule + time-embedding tables, computed on the host with the reference's expressions
        T = model.num_steps
        abar = model.alphas_cumprod.detach().float().cpu()
        betas = model.betas.detach().float().cpu()
        coef = torch.zeros(T, 6, dtype=torch.float32)
        for t in range(T):                       # 0-d tensor arithmetic, order of models/diffusion.py:401-419
            alpha_t = 1.0 - betas[t]
            ab = abar[t]
            coef[t, 0] = torch.sqrt(1 - ab)
            coef[t, 1] = torch.sqrt(ab)
            coef[t, 3] = 1 - ab
            if t > 0:
                abp = abar[t - 1]
                coef[t, 2] = torch.sqrt(abp) * betas[t]
                coef[t, 4] = torch.sqrt(alpha_t) * (1 - abp)
                coef[t, 5] = torch.sqrt((1 - abp) / (1 - ab) * betas[t])
        t_norm = torch.arange(T).float() / T     # == python t / T for every t < T
        temb = TimeEmbedding(model._time_dim)(t_norm).contiguous()
        sa = model.sqrt_alphas_cumprod.detach().float().cpu().contiguous()
        s1 = model.sqrt_one_minus_alphas_cumprod.detach().float().cpu().contiguous()
        L.check(lib.osd_set_schedule(self.handle, L.ptr(sa), L.ptr(s1), L.ptr(coef.contiguous()), L.ptr(temb)))
        self._sig = None
        self.constraints_version = 0
        self.serial = 0                 # bumped by every call that rewrites the handle's training workspace

    def set_constraints(self, spec):
        lib = L.lib()
        if spec is None:
            L.check(lib.osd_set_constraints(self.handle, None))
            return
        from .constraints import csr_from_pathways
        off, mem = csr_from_pathways(spec["pathways"])
        ca = np.ascontiguousarray(spec["cols_a"], dtype=np.int32)
        cb = np.ascontiguousarray(spec["cols_b"], dtype=np.int32)
        c = L.OsdConstraints()
        i32p = C.POINTER(C.c_int32)
        c.pathway_offsets, c.pathway_members, c.n_pathways = off.ctypes.data_as(i32p), mem.ctypes.data_as(i32p), len(off) - 1
        c.pathway_weight = spec["w_pc"]
        c.cols_a, c.cols_b, c.n_a, c.n_b = ca.ctypes.data_as(i32p), cb.ctypes.data_as(i32p), len(ca), len(cb)
        c.mutexpr_weight = spec["w_me"]
        L.check(lib.osd_set_constraints(self.handle, C.byref(c)))

    def close(self):
        if self.handle:
            L.lib().osd_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self, model: "BiologyAwareDiffusionModel"):
        """Bind the current stream; re-hand the parameter pointers if any tensor moved or changed."""
        lib = L.lib()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        L.check(lib.osd_set_stream(self.handle, C.c_void_p(stream)))
        params = model._param_list()
        sig = tuple((p.data_ptr(), p._version) for p in params)
        if sig != self._sig:
            for p in params:
                if p.dtype != torch.float32 or not p.is_contiguous() or p.device != self.device:
                    raise RuntimeError("parameters must be contiguous fp32 tensors on the model's device")
            arr = L.ptr_array(params)
            L.check(lib.osd_load_weights(self.handle, arr, len(params)))
            self._sig = sig


class BiologyAwareDiffusionModel(nn.Module):
    """Drop-in for models/diffusion.py:259 -- see module docstring."""

    def __init__(self, mutation_dim: int, expression_dim: int, pathway_dim: int, condition_dim: int, config: dict):
        super().__init__()
        self.mutation_dim = mutation_dim
        self.expression_dim = expression_dim
        self.pathway_dim = pathway_dim
        self.condition_dim = condition_dim
        self.data_dim = mutation_dim + expression_dim + pathway_dim

        m = config["model"]
        self._time_dim = int(m["latent_dim"])
        self._hidden_dims = [int(v) for v in m["hidden_dims"]]
        self._dropout_p = float(m["gnn"]["dropout"])          # the GNN key, as models/diffusion.py:294
        for h in self._hidden_dims:
            if h % 8:
                raise ValueError("num_channels must be divisible by num_groups")   # nn.GroupNorm's message
        if self._time_dim // 2 != COND_EMBED_WIDTH:
            raise ValueError("config.model.latent_dim // 2 must equal 64: the reference's ConditionalEmbedding "
                             "is 64 wide while cond_proj expects latent_dim // 2 inputs (models/diffusion.py:285,292)")

        self.condition_embed = ConditionalEmbedding(num_continuous=condition_dim, embedding_dim=COND_EMBED_WIDTH)
        self.unet = DiffusionUNet(data_dim=self.data_dim, time_dim=self._time_dim,
                                  condition_dim=self._time_dim // 2, hidden_dims=self._hidden_dims,
                                  dropout=self._dropout_p)

        self.num_steps = int(m["diffusion"]["num_steps"])
        self.register_buffer("betas", self._get_beta_schedule(m["diffusion"]["beta_schedule"], self.num_steps))
        alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.register_buffer("alphas_cumprod", alphas_cumprod)
        self.register_buffer("sqrt_alphas_cumprod", torch.sqrt(alphas_cumprod))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", torch.sqrt(1.0 - alphas_cumprod))
        self._engines = {}
        # sampling tunables forwarded to the library (rows per chunk, chunks in flight, hipGraph replay)
        self.sample_chunk_rows: Optional[int] = None
        self.sample_streams: Optional[int] = None
        self.use_graph: bool = True
        # reverse-chain engine: "auto" (library default: the persistent chain kernel for large eval-mode batches of a
        # 256/512-wide architecture, else the per-layer kernels), "chain", "graph" (per-layer kernels; hipGraph iff use_graph)
        self.sampler: str = "auto"
        # which chain kernel: "workspace" (csrc/chain.h: 128-row tiles, activations through a private workspace -- the faster one from
        # 65 536 rows on), "panel" (csrc/chain_panel.h: 64 patients per workgroup, activations in LDS; bit-identical; at its full
        # rate from 16 384 rows on; architectures whose panels do not fit run the workspace kernel), None / "auto" (the library's choice:
        # workspace from 65 536 rows on, panel from 10 240 rows on, squad up to 3 072 rows), "squad" (csrc/chain_squad.h: eight workgroups per
        # 32 patients, the small-batch kernel; agrees with the other engines to fp32 rounding, not bitwise; batches it cannot keep resident run
        # on auto's other choices)
        self.chain_variant: Optional[str] = None
        self.last_chain_variant: Optional[str] = None     # the one the most recent chain-kernel sample() used
        # patients per panel of the squad chain: None (the library's choice: 16 up to one 32-patient workgroup per CU -- ~1 000 rows --, else 32), 16, 32
        self.squad_panel: Optional[int] = None
        self.last_squad_panel: Optional[int] = None
        self.chain_grid: Optional[int] = None             # workgroup count of the chain kernel (tests)
        self.chain_steps_per_launch: Optional[int] = None
        self.chain_stagger: Optional[int] = None
        self.chain_spin_budget: Optional[int] = None      # ticks (100 MHz) a dependency wait inside the chain kernel may take
        self.chain_wall_budget_ms: Optional[int] = None   # host-side budget of a chain (0 / None: 10 x the estimate + 2 s)
        # per-layer engine, input_proj split-K over workgroups (small-batch latency path): None / 0 off -- a row's result is then
        # independent of chunking and sharding, bit for bit --, -1 auto (chunks with < 128 input_proj tiles), n = slices.
        # SyntheticPatientGenerator switches None to auto: its per-scenario batches are the reference's default workload
        self.input_splitk: Optional[int] = None
        self.last_sampler: Optional[str] = None           # engine the most recent sample() ran on
        # arithmetic of the eval-mode forward / p_sample / sample GEMMs: None / "fp32" = v_mfma_f32_32x32x2_f32, the reference's F.linear in
        # fp32 (models/diffusion.py:198-256; the default); "bf16x3" = every fp32 operand as three bf16 planes (exact) and six bf16 MFMAs
        # per product with fp32 accumulation (csrc/gemm_bf3.h): fp32 accuracy -- the same stated tolerances -- at the bf16 matrix rate.
        # Trunk widths 256 / 512 only; train-mode (dropout) calls and training stay fp32.
        self.precision: Optional[str] = None
        self.last_precision: Optional[str] = None         # what the most recent predict_noise / p_sample / sample computed in
        self.train_streams: Optional[int] = None      # 1 = whole backward on one stream, 2 (library default) = weight gradients on a side stream
        # the ten Linear+GroupNorm+SiLU layers of a training forward pass as one launch of squads (csrc/train_squad.h) from 2 048 rows on:
        # None / True (library default) or False (per-layer launches)
        self.train_squad = None       # None (library default) | False / 0 | True / 1 (forward only) | 2 (forward and the dgrad chain, csrc/train_squad_bwd.h)
        # the conditioning branch's backward below h0 (time-table scatter, two 64-wide dgrads, SiLU backward) as one launch
        # (k_cond_bwd, csrc/k_train.hip): None / True (library default) or False (four launches)
        self.cond_bwd_fused = None
        # optional constraint losses (set_constraints); None = the reference's eps-MSE only
        self._constraints = None
        self._constraints_version = 0

    # -- constraint losses (north_star; stubs at models/cvae.py:262-302) -----------------------------
    def set_constraints(self, pathways=None, mutation_columns=None, target_columns=None, *, pathway_weight: Optional[float] = None,
                        mutexpr_weight: Optional[float] = None, config: Optional[dict] = None):
        """Add the pathway-coherence and/or mutation-expression terms to the training loss (``forward``).

        pathways: list of member-column lists (columns of the D-wide feature vector; see
        ``constraints.pathways_from_matrix``); mutation_columns / target_columns: the two column sets of the
        correlation block (at most 64 each).  Weights default to ``config['model']['constraints']``
        (config.yaml:57-60) when a config is given, else 1.0.  Call with no arguments to clear."""
        cons = (config or {}).get("model", {}).get("constraints", {})
        if pathways is None and mutation_columns is None:
            self._constraints = None
        else:
            if (mutation_columns is None) != (target_columns is None):
                raise ValueError("mutation_columns and target_columns go together")
            self._constraints = {
                "pathways": [list(map(int, p)) for p in (pathways or [])],
                "cols_a": list(map(int, mutation_columns or [])), "cols_b": list(map(int, target_columns or [])),
                "w_pc": float(pathway_weight if pathway_weight is not None else cons.get("pathway_coherence_weight", 1.0)),
                "w_me": float(mutexpr_weight if mutexpr_weight is not None else cons.get("mutation_expression_weight", 1.0)),
            }
        self._constraints_version += 1

    def last_loss_parts(self):
        """(mse, L_pc, L_me) of the most recent training ``forward`` with constraints configured."""
        eng = self._engine()
        out = (C.c_float * 3)()
        L.check(L.lib().osd_get_loss_parts(eng.handle, out))
        return float(out[0]), float(out[1]), float(out[2])

    # -- schedule: same torch expressions as models/diffusion.py:312-326, hence bit-identical buffers
    def _get_beta_schedule(self, schedule_type: str, num_steps: int):
        if schedule_type == "linear":
            return torch.linspace(1e-4, 0.02, num_steps)
        if schedule_type == "cosine":
            steps = torch.arange(num_steps + 1, dtype=torch.float32) / num_steps
            abar = torch.cos((steps + 0.008) / 1.008 * np.pi / 2) ** 2
            abar = abar / abar[0]
            return torch.clip(1 - (abar[1:] / abar[:-1]), 0.0001, 0.9999)
        raise ValueError(f"Unknown schedule: {schedule_type}")

    # -- plumbing ---------------------------------------------------------------------------
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engines"] = {}          # device handles are per-process; rebuilt lazily
        return state

    def __deepcopy__(self, memo):
        import copy
        engines, self._engines = self._engines, {}
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                setattr(new, k, copy.deepcopy(v, memo))
        finally:
            self._engines = engines
        return new

    def _param_list(self):
        return list(self.parameters())

    def _device(self) -> torch.device:
        return next(self.parameters()).device

    def _engine(self) -> _Engine:
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("BiologyAwareDiffusionModel (osteosarcoma_diffusionmodel_amd) runs on MI355X only: "
                               "move the model to a ROCm device (model.to('cuda')); there is no CPU fallback")
        key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
        eng = self._engines.get(key)
        if eng is None:
            with torch.cuda.device(dev):
                eng = _Engine(self, torch.device("cuda", key[1]))
            self._engines[key] = eng
        eng.sync(self)
        if eng.constraints_version != self._constraints_version:
            eng.set_constraints(self._constraints)
            eng.constraints_version = self._constraints_version
        if self.sample_chunk_rows:
            L.check(L.lib().osd_set_option(eng.handle, b"chunk_rows", int(self.sample_chunk_rows)))
        if self.sample_streams:
            L.check(L.lib().osd_set_option(eng.handle, b"n_streams", int(self.sample_streams)))
        if self.train_streams:
            L.check(L.lib().osd_set_option(eng.handle, b"train_streams", int(self.train_streams)))
        if self.train_squad is not None:
            L.check(L.lib().osd_set_option(eng.handle, b"train_squad", int(self.train_squad)))
        if self.cond_bwd_fused is not None:
            L.check(L.lib().osd_set_option(eng.handle, b"cond_bwd_fused", int(bool(self.cond_bwd_fused))))
        try:
            mode = {"auto": 0, "chain": 1, "graph": 2, "layers": 2}[self.sampler]
        except KeyError:
            raise ValueError(f"sampler must be 'auto', 'chain' or 'graph', got {self.sampler!r}")
        L.check(L.lib().osd_set_option(eng.handle, b"sampler", mode))
        try:
            variant = {None: 0, "auto": 0, "workspace": 1, "panel": 2, "squad": 3}[self.chain_variant]
        except KeyError:
            raise ValueError(f"chain_variant must be None, 'auto', 'workspace', 'panel' or 'squad', got {self.chain_variant!r}")
        L.check(L.lib().osd_set_option(eng.handle, b"chain_variant", variant))
        if self.squad_panel not in (None, 0, 16, 32):
            raise ValueError(f"squad_panel must be None, 16 or 32, got {self.squad_panel!r}")
        L.check(L.lib().osd_set_option(eng.handle, b"squad_panel", int(self.squad_panel or 0)))
        try:
            prec = {None: 0, "fp32": 0, "f32": 0, "bf16x3": 1}[self.precision]
        except KeyError:
            raise ValueError(f"precision must be None, 'fp32' or 'bf16x3', got {self.precision!r}")
        L.check(L.lib().osd_set_option(eng.handle, b"precision", prec))
        for name, val in (("chain_grid", self.chain_grid), ("chain_steps_per_launch", self.chain_steps_per_launch),
                          ("chain_stagger", self.chain_stagger), ("chain_spin_budget", self.chain_spin_budget),
                          ("chain_wall_budget_ms", self.chain_wall_budget_ms), ("input_splitk", self.input_splitk)):
            if val is not None:
                L.check(L.lib().osd_set_option(eng.handle, name.encode(), int(val)))
        return eng

    def _prep(self, t: torch.Tensor, cols: Optional[int] = None, name: str = "tensor") -> torch.Tensor:
        dev = self._device()
        if t.device != dev:
            raise RuntimeError(f"{name} is on {t.device} but the model is on {dev}")
        t = t.to(torch.float32).contiguous()
        if cols is not None and (t.dim() != 2 or t.shape[1] != cols):
            raise RuntimeError(f"{name}: expected shape [N, {cols}], got {tuple(t.shape)}")
        return t

    def _t32(self, t: torch.Tensor, n: int, device) -> torch.Tensor:
        """Caller-supplied per-row timestep indices as device int32, range-checked on the host: the reference's
        buffer gather (models/diffusion.py:337) raises IndexError for t outside [0, T).  Costs one sync, only on
        the injected-t path (the default path draws t on the device)."""
        t32 = t.to(device=device, dtype=torch.int32).contiguous()
        if t32.numel() != n:
            raise RuntimeError("t must have one entry per row")
        if n and (int(t32.min()) < 0 or int(t32.max()) >= self.num_steps):
            raise IndexError(f"timestep index out of range [0, {self.num_steps})")
        return t32

    def _note_precision(self, eng) -> None:
        v = C.c_int64(0)
        L.check(L.lib().osd_get_option(eng.handle, b"last_precision", C.byref(v)))
        self.last_precision = "bf16x3" if int(v.value) == 1 else "fp32"

    def _flags(self) -> int:
        return L.OSD_F_TRAIN_MODE if self.training else 0

    # -- q_sample (models/diffusion.py:328-342) -------------------------------------------------
    def q_sample(self, x_0, t, noise=None, *, seed: Optional[int] = None):
        eng = self._engine()
        x_0 = self._prep(x_0, self.data_dim, "x_0")
        n = x_0.shape[0]
        t32 = self._t32(t, n, x_0.device)
        x_t = torch.empty_like(x_0)
        if noise is None:
            noise_out = torch.empty_like(x_0)
            seed = _draw_seed() if seed is None else seed
            L.check(L.lib().osd_q_sample(eng.handle, L.ptr(x_0), L.ptr(t32), None, n, seed, 0, L.ptr(x_t), L.ptr(noise_out)))
            return x_t, noise_out
        noise = self._prep(noise, self.data_dim, "noise")
        L.check(L.lib().osd_q_sample(eng.handle, L.ptr(x_0), L.ptr(t32), L.ptr(noise), n, 0, 0, L.ptr(x_t), None))
        return x_t, noise

    # -- training forward (models/diffusion.py:344-380) ------------------------------------------
    def forward(self, x_0, conditions, return_loss=True, *, t=None, noise=None, dropout_masks=None,
                seed: Optional[int] = None):
        from .train import diffusion_loss   # autograd.Function around osd_train_loss_fwd_bwd
        if return_loss:
            return diffusion_loss(self, x_0, conditions, t=t, noise=noise, dropout_masks=dropout_masks, seed=seed)
        # return_loss=False: predicted noise for freshly drawn (or injected) t / noise
        eng = self._engine()
        x_0 = self._prep(x_0, self.data_dim, "x_0")
        conditions = self._prep(conditions, self.condition_dim, "conditions")
        n = x_0.shape[0]
        seed = _draw_seed() if seed is None else seed
        if t is None:
            t = torch.randint(0, self.num_steps, (n,), device=x_0.device)
        x_t, _ = self.q_sample(x_0, t, noise, seed=seed)
        return self.predict_noise(x_t, t, conditions, dropout_masks=dropout_masks, seed=seed)

    def predict_noise(self, x_t, t, conditions, *, dropout_masks: Optional[Sequence[torch.Tensor]] = None,
                      seed: Optional[int] = None):
        """DiffusionUNet.forward(x_t, t/T, condition_embed(c)) (models/diffusion.py:370-373); ``t`` is an
        int (shared) or an integer tensor of per-row timestep indices."""
        eng = self._engine()
        x_t = self._prep(x_t, self.data_dim, "x_t")
        conditions = self._prep(conditions, self.condition_dim, "conditions")
        n = x_t.shape[0]
        eps = torch.empty_like(x_t)
        if isinstance(t, int):
            t_idx, t_all = None, t
        else:
            t_idx, t_all = self._t32(t, n, x_t.device), 0
        flags = self._flags()
        masks = None
        if dropout_masks is not None:
            flags |= L.OSD_F_TRAIN_MODE
            keep = [self._prep(m, name="dropout mask") for m in dropout_masks]
            masks = L.ptr_array(keep)
        seed = _draw_seed() if seed is None else seed
        if torch.is_grad_enabled() and (x_t.requires_grad or any(p.requires_grad for p in self.parameters())):
            # differentiable path (custom losses): activations kept for osd_denoiser_backward
            from .train import denoiser_with_grad
            if t_idx is None:
                t_idx = torch.full((n,), t_all, device=x_t.device, dtype=torch.int32)
            self.last_precision = "fp32"          # the differentiable path keeps the activations of the fp32 kernels
            return denoiser_with_grad(self, x_t, t_idx, conditions, keep if dropout_masks is not None else None, seed, flags)
        L.check(L.lib().osd_denoiser_forward(eng.handle, L.ptr(x_t), L.ptr(t_idx), t_all, L.ptr(conditions), n,
                                             L.ptr(eps), flags, masks, seed))
        self._note_precision(eng)
        return eps

    # -- p_sample / sample (models/diffusion.py:382-449) ------------------------------------------
    @torch.no_grad()
    def p_sample(self, x_t, t, conditions, *, noise=None, seed: Optional[int] = None):
        eng = self._engine()
        x_t = self._prep(x_t, self.data_dim, "x_t")
        conditions = self._prep(conditions, self.condition_dim, "conditions")
        n = x_t.shape[0]
        out = torch.empty_like(x_t)
        z = None if noise is None else self._prep(noise, self.data_dim, "noise")
        seed = _draw_seed() if (seed is None and z is None) else (seed or 0)
        L.check(L.lib().osd_p_sample_step(eng.handle, L.ptr(x_t), int(t), L.ptr(conditions), L.ptr(z), n, seed, 0,
                                          L.ptr(out), self._flags()))
        self._note_precision(eng)
        return out

    @torch.no_grad()
    def sample(self, conditions, num_samples: int = 1, *, x_T=None, noise=None, seed: Optional[int] = None,
               row_offset: int = 0, return_mutation_mask: bool = False):
        """Full reverse chain.  ``x_T`` [N,D] and ``noise`` [T-1,N,D] (draw order t = T-1..1) inject the
        random draws; otherwise Philox(seed, row_offset + row) generates them on the device."""
        eng = self._engine()
        conditions = self._prep(conditions, self.condition_dim, "conditions")
        n = int(num_samples)
        if conditions.shape[0] != n:
            # the reference broadcasts-or-fails here (models/diffusion.py:443-447)
            raise RuntimeError(f"conditions has {conditions.shape[0]} rows but num_samples is {n}")
        out = torch.empty(n, self.data_dim, device=conditions.device, dtype=torch.float32)
        xT = None if x_T is None else self._prep(x_T, self.data_dim, "x_T")
        zs = None
        if noise is not None:
            zs = noise.to(torch.float32).contiguous()
            if tuple(zs.shape) != (self.num_steps - 1, n, self.data_dim):
                raise RuntimeError(f"noise: expected shape [{self.num_steps - 1}, {n}, {self.data_dim}]")
        mask = torch.empty(n, self.mutation_dim, device=out.device, dtype=torch.float32) if return_mutation_mask else None
        if seed is None:
            seed = _draw_seed()
        flags = self._flags() | (L.OSD_F_GRAPH if self.use_graph else 0)
        engine = L.lib().osd_sample_engine(eng.handle, n, flags)
        if engine < 0:
            L.check(engine)
        if engine == 1:
            # the chain kernel's bounded waits report through a status word: the synchronous call reads it and, should the
            # chain have given up, re-runs it on the per-layer kernels (same bits) -- sample() cannot fail, as the reference's
            flags |= L.OSD_F_SYNC

        def counter(name):
            v = C.c_int64(0)
            L.check(L.lib().osd_get_option(eng.handle, name, C.byref(v)))
            return int(v.value)

        gave_up_before = counter(b"chain_fallbacks")
        L.check(L.lib().osd_sample_chain(eng.handle, L.ptr(conditions), n, L.ptr(xT), L.ptr(zs), seed, int(row_offset),
                                         L.ptr(out), L.ptr(mask), flags))
        used = L.lib().osd_sample_engine(eng.handle, -1, 0)      # the engine that produced the result
        if used < 0:
            L.check(used)
        self.last_sampler = "chain" if used == 1 else "graph"
        self._note_precision(eng)
        # a chain kernel ran iff the result is its own or it gave up and was re-run (the counter moved); a call the library
        # demoted up front (injected draws at D % 4 != 0 keep the guarded per-layer kernels) launched none and warns about nothing
        gave_up = counter(b"chain_fallbacks") > gave_up_before
        self.last_chain_variant = self.last_squad_panel = None
        if used == 1 or gave_up:
            self.last_chain_variant = {1: "workspace", 2: "panel", 3: "squad"}.get(counter(b"last_chain_variant"))
            self.last_squad_panel = counter(b"last_squad_panel") if self.last_chain_variant == "squad" else None
        if gave_up:
            import warnings
            warnings.warn(L.last_error() or "the reverse-chain kernel gave up; the chain was re-run on the per-layer kernels")
        if return_mutation_mask:
            return out, mask
        return out


# north_star alias (there is no class of this name in the reference)
BiologyAwareDiffusion = BiologyAwareDiffusionModel
