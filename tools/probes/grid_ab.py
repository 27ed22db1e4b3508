import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from bench import CONF, scenario_conditions
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
n, steps = 100000, 60
conf = {"model": dict(CONF["model"])}; conf["model"]["diffusion"] = {"num_steps": steps, "beta_schedule": "cosine"}
torch.manual_seed(0)
cond = scenario_conditions(n, 0).cuda()
for grid in (512, 256, 384, 512, 256):
    m = BiologyAwareDiffusionModel(50, 1900, 50, 3, conf).cuda().eval()
    m.sampler, m.chain_grid = "chain", grid
    m.sample(cond, n, seed=1); torch.cuda.synchronize()
    t0 = time.perf_counter(); m.sample(cond, n, seed=2); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"grid {grid}: {dt*1e3:.1f} ms  {n*steps/dt/1e6:.2f} M patient-steps/s", flush=True)
