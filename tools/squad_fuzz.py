#!/usr/bin/env python3
"""One-off robustness check: the squad chain kernel against the per-layer kernels at the chain tolerance, and against itself bit for
bit, over random row counts (1 .. 3 072: one, two and three workgroups per CU), feature counts (any alignment; 8 .. 190 state tiles,
i.e. every left-over-tile case), mutation widths, launch segmentation, injected and Philox draws."""
import sys, random, torch
from pathlib import Path; R = Path(__file__).resolve().parent.parent; sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
from helpers import FULL_H, config
random.seed(11)
bad = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 32):
    T = random.choice([1, 2, 3, 6])
    n = random.choice([1, 2, 31, 32, 33, 64, 200, 999, 1023, 1025, 2048, 2049, 3072, random.randint(1, 3072)])
    spl = random.choice([0, 0, 1, 2])
    D = random.choice([2000, 5142, 256, 257, 288, 511, 502, 1000, 1025, 3333, random.randint(226, 6080)])
    mut = random.choice([1, 7, 50, 62, min(200, D - 40)])
    inject = random.random() < 0.4 and T > 1 and n * D * T < 2e8
    dims = dict(mutation_dim=mut, expression_dim=D - mut - 10, pathway_dim=10, condition_dim=3)
    torch.manual_seed(case)
    m = BiologyAwareDiffusionModel(config=config(FULL_H, T=T), **dims).cuda().eval()
    m.input_splitk = 0
    g = torch.Generator().manual_seed(case)
    cond = torch.randn(n, 3, generator=g).cuda()
    kw = dict(seed=case, row_offset=case * 3)
    if inject:
        kw = dict(x_T=torch.randn(n, D, generator=g).cuda(), noise=torch.randn(T - 1, n, D, generator=g).cuda())
    m.sampler, m.chain_variant = "graph", None
    ref, refm = m.sample(cond, n, return_mutation_mask=True, **kw)
    m.sampler, m.chain_variant = "chain", "squad"
    out, mask = m.sample(cond, n, return_mutation_mask=True, **kw)
    ran = (m.last_sampler, m.last_chain_variant)
    if spl: m.chain_steps_per_launch = spl
    out2, mask2 = m.sample(cond, n, return_mutation_mask=True, **kw)
    scale = ref.abs().max().item()
    d = (out - ref).abs().max().item()
    near = (ref[:, :mut] - 0.5).abs() <= 2e-5 * scale + 1e-6
    ok = ran == ("chain", "squad") and bool(torch.isfinite(out).all()) and d <= 2e-5 * scale + 1e-6 and ((mask != refm) & ~near).sum().item() == 0 \
        and torch.equal(out2, out) and torch.equal(mask2, mask)
    bad += not ok
    print(f"case {case}: T={T} n={n} seg={spl} D={D} mut={mut} {'injected' if inject else 'philox'} -> {'ok' if ok else 'MISMATCH'} {ran} max|d|/max|ref| = {d / max(scale, 1e-30):.2e}", flush=True)
print("mismatches:", bad)
