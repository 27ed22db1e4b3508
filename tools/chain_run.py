#!/usr/bin/env python3
"""Run the reverse chain once at a chosen T (diagnostic target of rocprofv3 passes): chain_run.py <rows> <T> <sampler|split> [workspace|panel|squad].
`split` = the bf16x3 precision (per-layer launches on planes buffers, csrc/split.hip)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import CONF, scenario_conditions
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
conf = {"model": dict(CONF["model"])}; conf["model"]["diffusion"] = {"num_steps": steps, "beta_schedule": "cosine"}
torch.manual_seed(0)
m = BiologyAwareDiffusionModel(50, 1900, 50, 3, conf).cuda().eval()
m.sampler = sys.argv[3] if len(sys.argv) > 3 else "chain"
if m.sampler == "split":
    m.sampler, m.precision = "auto", "bf16x3"
m.chain_variant = sys.argv[4] if len(sys.argv) > 4 else "workspace"
cond = scenario_conditions(n, 0).cuda()
for i in range(2):
    out = m.sample(cond, n, seed=1 + i)
torch.cuda.synchronize()
print("ok", m.last_sampler, m.last_chain_variant, m.last_precision, float(out.abs().max()))
