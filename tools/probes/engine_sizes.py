#!/usr/bin/env python3
"""Probe: patient-steps/s of the three reverse-chain engines (per-layer kernels, workspace chain, LDS-resident chain) over batch sizes."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
from tests.helpers import FULL, FULL_H, config
T = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.manual_seed(0)
m = BiologyAwareDiffusionModel(config=config(FULL_H, T=T), **FULL).cuda().eval()
m.input_splitk = 0
sizes = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else (8192, 16384, 24576, 32768, 49152, 65536, 100000, 131072)
for n in sizes:
    cond = torch.randn(n, 3, generator=torch.Generator().manual_seed(3)).cuda()
    row = []
    for name, sampler, variant in (("per-layer", "graph", None), ("workspace", "chain", "workspace"), ("panel", "chain", "panel")):
        m.sampler, m.chain_variant = sampler, variant
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.sample(cond, n, seed=5)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        row.append(f"{name} {n * T / best / 1e6:6.2f}")
    print(f"n={n:7d}: " + "  ".join(row) + "   M patient-steps/s", flush=True)
