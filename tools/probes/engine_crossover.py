#!/usr/bin/env python3
"""Which reverse-chain engine is faster at which row count?  (tunes chain_pick_engine's threshold)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from bench import CONF, scenario_conditions
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
T = 60
conf = {"model": dict(CONF["model"])}; conf["model"]["diffusion"] = {"num_steps": T, "beta_schedule": "cosine"}
torch.manual_seed(0)
m = BiologyAwareDiffusionModel(50, 1900, 50, 3, conf).cuda().eval()
for n in (4096, 8192, 16384, 24576, 32768, 49152, 65536):
    cond = scenario_conditions(n, 0).cuda()
    res = {}
    for eng in ("chain", "graph"):
        m.sampler = eng
        m.sample(cond, n, seed=1); torch.cuda.synchronize()
        t0 = time.perf_counter(); m.sample(cond, n, seed=2); torch.cuda.synchronize()
        res[eng] = n * T / (time.perf_counter() - t0) / 1e6
    print(f"n={n:6d}: chain {res['chain']:6.2f}  per-layer {res['graph']:6.2f}  M patient-steps/s", flush=True)
