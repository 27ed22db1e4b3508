#!/usr/bin/env python3
"""Small-batch generation (the reference's default sizes: 3 x 333 / 3 x 1000 patients at dims 62 / 5054 / 26): time per reverse step of
every engine.   small_batch.py <rows> <T> [engine ...]      engines: graph (fp32 per-layer), split (bf16x3), auto"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import CONF, scenario_conditions
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
engines = sys.argv[3:] or ["graph", "split"]
conf = {"model": dict(CONF["model"])}; conf["model"]["diffusion"] = {"num_steps": steps, "beta_schedule": "cosine"}
torch.manual_seed(0)
m = BiologyAwareDiffusionModel(62, 5054, 26, 3, conf).cuda().eval()
m.input_splitk = -1                # SyntheticPatientGenerator's default for its own model
cond = scenario_conditions(n, 0).cuda()
for e in engines:
    m.precision = "bf16x3" if e == "split" else None
    m.sampler = "auto" if e in ("split", "auto") else e
    m.sample(cond, n, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.sample(cond, n, seed=2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{e:6s} rows {n} D 5142: {dt / steps * 1e6:8.1f} us per step = {n / (dt / steps * 1000):8.0f} patients/s at T = 1000   [{m.last_sampler}/{m.last_precision}]", flush=True)
