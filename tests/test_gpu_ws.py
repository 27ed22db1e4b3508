"""GPU: the wave-specialised persistent GEMM kernel (csrc/gemm_ws.h) must be bit-identical to the one-tile-per-workgroup
kernel it replaces: same K order per accumulator, same epilogue code."""
import ctypes as C

import pytest
import torch

from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, _lib as L
from helpers import FULL, FULL_H, config

pytestmark = pytest.mark.gpu


def _set_ws(model, v):
    L.check(L.lib().osd_set_option(model._engine().handle, b"wave_specialized", v))


@pytest.mark.parametrize("rows", [33000, 4097])
def test_sampling_chain_bitwise_equal(rows):
    torch.manual_seed(0)
    m = BiologyAwareDiffusionModel(config=config(FULL_H, T=12), **FULL).cuda().eval()
    m.sample_chunk_rows = 1 << 20
    cond = torch.randn(rows, 3, device="cuda")
    outs = []
    try:
        for ws in (0, 2, 0, 2):
            _set_ws(m, ws)
            x, mask = m.sample(cond, rows, seed=77, return_mutation_mask=True)
            outs.append((x.clone(), mask.clone()))
    finally:
        _set_ws(m, 0)
    assert torch.isfinite(outs[0][0]).all()
    for x, mk in outs[1:]:
        assert torch.equal(x, outs[0][0]) and torch.equal(mk, outs[0][1])


def test_single_step_and_eager_forward_bitwise_equal():
    torch.manual_seed(1)
    m = BiologyAwareDiffusionModel(config=config(FULL_H, T=1000), **FULL).cuda().eval()
    rows = 40000
    x = torch.randn(rows, 2000, device="cuda")
    cond = torch.randn(rows, 3, device="cuda")
    try:
        with torch.no_grad():
            _set_ws(m, 0)
            a = m.predict_noise(x, 500, cond)
            pa = m.p_sample(x, 500, cond, seed=5)
            _set_ws(m, 1)
            b = m.predict_noise(x, 500, cond)
            pb = m.p_sample(x, 500, cond, seed=5)
    finally:
        _set_ws(m, 0)
    assert torch.equal(a, b) and torch.equal(pa, pb)
