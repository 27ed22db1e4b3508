#!/usr/bin/env python3
"""First contact of the LDS-resident chain kernel: bits vs the per-layer engine, then a timing of both chain kernels."""
import sys, time
import torch
sys.path.insert(0, ".")
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
from tests.helpers import FULL, FULL_H, config

def model(T, seed=0):
    torch.manual_seed(seed)
    m = BiologyAwareDiffusionModel(config=config(FULL_H, T=T), **FULL).cuda().eval()
    m.input_splitk = 0
    return m

T = 8
m = model(T)
for n, grid in ((64, 1), (130, 2), (1000, 3), (5000, 0)):
    cond = torch.randn(n, 3, generator=torch.Generator().manual_seed(3)).cuda()
    m.sampler = "graph"
    ref, refm = m.sample(cond, n, return_mutation_mask=True, seed=77, row_offset=5)
    for variant in ("workspace", "panel"):
        m.sampler, m.chain_variant, m.chain_grid = "chain", variant, grid
        out, mask = m.sample(cond, n, return_mutation_mask=True, seed=77, row_offset=5)
        d = (out - ref).abs().max().item()
        print(f"n={n} grid={grid} {variant:9s} ran={m.last_sampler}/{m.last_chain_variant} finite={bool(torch.isfinite(out).all())} "
              f"max|d|={d:.3e} of {ref.abs().max().item():.3e} bitwise={torch.equal(out, ref)} mask={torch.equal(mask, refm)}", flush=True)

if len(sys.argv) > 1:
    n, T = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 50
    m = model(T)
    m.chain_grid = None
    cond = torch.randn(n, 3, generator=torch.Generator().manual_seed(3)).cuda()
    for variant in ("workspace", "panel") * (int(sys.argv[3]) if len(sys.argv) > 3 else 2):
        m.sampler, m.chain_variant = "chain", variant
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = m.sample(cond, n, seed=5)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        flop = 5193728.0 * n * T
        print(f"{variant:9s} n={n} T={T}: {dt*1e3:8.1f} ms  {n*T/dt/1e6:7.2f} M patient-steps/s  {flop/dt/1e12:6.1f} TFLOP/s ({flop/dt/1e12/157.3:.3f})", flush=True)
