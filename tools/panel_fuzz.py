#!/usr/bin/env python3
"""One-off robustness check: the LDS-resident chain kernel against the per-layer kernels, bit for bit, over random row counts,
workgroup caps, launch segmentation and feature counts (padded and unpadded states)."""
import sys, random, torch
from pathlib import Path; R = Path(__file__).resolve().parent.parent; sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
from helpers import FULL, FULL_H, config
random.seed(7)
bad = 0
for case in range(24):
    T = random.choice([1, 2, 3, 7])
    n = random.choice([1, 2, 63, 64, 65, 127, 129, 200, 777, 2049, random.randint(1, 3000)])
    grid = random.choice([0, 1, 2, 3, 5, 300])
    spl = random.choice([0, 0, 1, 2])
    D = random.choice([2000, 2000, 512, 516, 1028, 2052, 700])
    dims = dict(mutation_dim=50, expression_dim=D - 60, pathway_dim=10, condition_dim=3)
    torch.manual_seed(case)
    m = BiologyAwareDiffusionModel(config=config(FULL_H, T=T), **dims).cuda().eval()
    m.input_splitk = 0
    cond = torch.randn(n, 3, generator=torch.Generator().manual_seed(case)).cuda()
    m.sampler = "graph"
    ref, refm = m.sample(cond, n, return_mutation_mask=True, seed=case, row_offset=case * 3)
    m.sampler, m.chain_variant, m.chain_grid = "chain", "panel", grid
    if spl: m.chain_steps_per_launch = spl
    out, mask = m.sample(cond, n, return_mutation_mask=True, seed=case, row_offset=case * 3)
    ok = torch.equal(out, ref) and torch.equal(mask, refm) and m.last_chain_variant == "panel"
    bad += not ok
    print(f"case {case}: T={T} n={n} grid={grid} seg={spl} D={D} -> {'ok' if ok else 'MISMATCH'} ({m.last_sampler}/{m.last_chain_variant})", flush=True)
print("mismatches:", bad)
