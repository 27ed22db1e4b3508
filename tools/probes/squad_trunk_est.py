import sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
from helpers import FULL_H, config
T = 200
for n in (999, 2048, 3072):
    torch.manual_seed(0)
    m = BiologyAwareDiffusionModel(config=config(FULL_H, T=T), mutation_dim=3, expression_dim=250, pathway_dim=3, condition_dim=3).cuda().eval()
    cond = torch.randn(n, 3).cuda()
    for panel in (32,):
        m.sampler, m.chain_variant, m.squad_panel = "chain", "squad", panel
        m.sample(cond, n, seed=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.sample(cond, n, seed=2)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"D=256 n={n} panel={m.last_squad_panel}: {dt/T*1e6:.1f} us/step (trunk-dominated)", flush=True)
