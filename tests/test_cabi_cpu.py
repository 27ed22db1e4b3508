"""CPU: the C-ABI library loads and exports every symbol include/osdiff.h declares; host-side
argument validation works without a GPU (no compute calls)."""
import ctypes as C
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "osdiff.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(osd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from osteosarcoma_diffusionmodel_amd import _lib as L
    lib = L.lib()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"libosdiff.so does not export {s}"
    assert sorted(L.exported_symbols()) == syms        # the ctypes table covers the whole header
    assert lib.osd_version() == 100


def test_library_exports_nothing_undeclared():
    """The product .so carries exactly the header's entry points: diagnostics (csrc/diag, `make DIAG=1`) and
    experiments are not part of it."""
    import subprocess
    from osteosarcoma_diffusionmodel_amd import _lib as L
    out = subprocess.run(["nm", "-D", "--defined-only", str(L.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in "TW"
                       and ln.split()[-1].startswith("osd_")})
    assert exported == declared_symbols()


def test_arch_queries_without_gpu():
    from osteosarcoma_diffusionmodel_amd import _lib as L
    lib = L.lib()
    cfg = L.OsdConfig()
    cfg.mutation_dim, cfg.expression_dim, cfg.pathway_dim, cfg.condition_dim = 50, 1900, 50, 3
    cfg.time_dim, cfg.n_hidden = 128, 3
    for i, v in enumerate((256, 512, 256)):
        cfg.hidden_dims[i] = v
    cfg.num_steps, cfg.dropout_p = 1000, 0.2
    assert lib.osd_num_params(C.byref(cfg)) == 52
    assert sum(lib.osd_param_numel(C.byref(cfg), i) for i in range(52)) == 2663952
    cfg.hidden_dims[1] = 500                                                             # not divisible by 8
    assert lib.osd_num_params(C.byref(cfg)) == 0
    cfg.hidden_dims[1] = 512
    cfg.time_dim = 64                                                                    # latent_dim//2 != 64
    assert lib.osd_num_params(C.byref(cfg)) == 0


def test_parameter_order_matches_named_parameters():
    from oracle import diffusion_oracle as O
    from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, _lib as L
    conf = {"model": {"latent_dim": 128, "hidden_dims": [64, 128, 256, 64], "gnn": {"dropout": 0.1},
                      "diffusion": {"num_steps": 10, "beta_schedule": "linear"}}}
    m = BiologyAwareDiffusionModel(6, 30, 4, 5, conf)
    shapes = O.param_shapes(6, 30, 4, 5, [64, 128, 256, 64], 128)
    assert [k for k, _ in m.named_parameters()] == list(shapes)
    assert [tuple(p.shape) for p in m.parameters()] == list(shapes.values())
    cfg = L.OsdConfig()
    cfg.mutation_dim, cfg.expression_dim, cfg.pathway_dim, cfg.condition_dim = 6, 30, 4, 5
    cfg.time_dim, cfg.n_hidden = 128, 4
    for i, v in enumerate((64, 128, 256, 64)):
        cfg.hidden_dims[i] = v
    cfg.num_steps = 10
    lib = L.lib()
    assert lib.osd_num_params(C.byref(cfg)) == len(shapes)
    for i, p in enumerate(m.parameters()):
        assert lib.osd_param_numel(C.byref(cfg), i) == p.numel()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: without a device the model refuses to run instead of computing on the host."""
    from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, _lib as L
    conf = {"model": {"latent_dim": 128, "hidden_dims": [32, 64, 32], "gnn": {"dropout": 0.2},
                      "diffusion": {"num_steps": 10, "beta_schedule": "cosine"}}}
    m = BiologyAwareDiffusionModel(8, 24, 8, 3, conf)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.sample(torch.zeros(2, 3), 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 40), torch.zeros(2, 3))
    cfg = L.OsdConfig()
    cfg.mutation_dim, cfg.expression_dim, cfg.pathway_dim, cfg.condition_dim = 8, 24, 8, 3
    cfg.time_dim, cfg.n_hidden, cfg.num_steps = 128, 1, 10
    cfg.hidden_dims[0] = 32
    h = C.c_void_p()
    assert L.lib().osd_create(C.byref(cfg), C.byref(h)) == L.OSD_EHIP
    assert b"no CPU fallback" in L.lib().osd_last_error() or b"HIP" in L.lib().osd_last_error()
