#!/usr/bin/env python3
"""Diagnostic: cProfile of BiologicalValidator.validate_all at BASELINE config 5's one-GPU share (random data of that shape)."""
import cProfile, pstats, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from osteosarcoma_diffusionmodel_amd.validation import BiologicalValidator, DeviceFrame
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
nr = int(sys.argv[2]) if len(sys.argv) > 2 else n
md, ed, pd_ = 50, 1900, 50
mut_cols = [f"GENE_{i}" for i in range(md)]; expr_cols = [f"EXPR_{i}" for i in range(ed)]; path_cols = [f"PATHWAY_{i}" for i in range(pd_)]
val = BiologicalValidator({"evaluation": {"driver_genes": mut_cols[:4], "mutually_exclusive_pairs": [[mut_cols[0], mut_cols[1]]],
                                          "mutation_expression_rules": [{"mutation": "GENE_0", "pathway": "PATHWAY_0", "direction": "positive"}]}})
def frames(rows, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(rows, md + ed + pd_, device="cuda", generator=g)
    x[:, :md] = (x[:, :md] > 0.8).float()
    return (DeviceFrame(x[:, :md].contiguous(), mut_cols), DeviceFrame(x[:, md:md + ed].contiguous(), expr_cols), DeviceFrame(x[:, md + ed:].contiguous(), path_cols))
import pandas as pd
pgm = pd.DataFrame(0, index=expr_cols, columns=[f"HALLMARK_{i}" for i in range(10)])
for p in range(10): pgm.iloc[64 * p: 64 * p + 64, p] = 1
if len(sys.argv) > 3 and sys.argv[3] == "model":
    # bench.py's validate leg: "real" = scenario 3 samples, "synthetic" = scenario 1 samples of a random-init model (heavy tails)
    from bench import CONF, SCENARIOS
    from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
    torch.manual_seed(0)
    model = BiologyAwareDiffusionModel(50, 1900, 50, 3, CONF).cuda().eval()
    rows = [[(sc["survival_time"] - 800) / 500, sc["event_occurred"], sc["metastasis_at_diagnosis"]] for sc in SCENARIOS]
    def sample(seed, row, m):
        cond = torch.tensor([row], dtype=torch.float32, device="cuda").repeat(m, 1)
        x, mask = model.sample(cond, m, seed=seed, return_mutation_mask=True)
        x = torch.nan_to_num(x, nan=0.0, posinf=1e6, neginf=-1e6)
        x[:, :md] = mask
        return (DeviceFrame(x[:, :md].contiguous(), mut_cols), DeviceFrame(x[:, md:md + ed].contiguous(), expr_cols), DeviceFrame(x[:, md + ed:].contiguous(), path_cols))
    r = sample(777, rows[2], nr); s = sample(2000, rows[0], n)
else:
    r = frames(nr, 1); s = frames(n, 2)
val.compute_mmd(torch.cat([f.values for f in r], 1)[:2048], torch.cat([f.values for f in s], 1)[:2048])
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    out = val.validate_all(*r, *s, pathway_gene_matrix=pgm)
    pr.disable(); torch.cuda.synchronize()
    print(f"validate_all({nr} real, {n} synthetic): {time.perf_counter() - t0:.2f} s; wasserstein {out['wasserstein_distance_mean']:.5f} mmd {out['mmd']:.5f}", flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
