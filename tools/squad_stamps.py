#!/usr/bin/env python3
"""Diagnostic (needs `make DIAG=1`, OSDIFF_LIB=<that library>): where the cycles of the squad chain kernel's waves go.
   squad_stamps.py [rows T]"""
import ctypes as C, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import CONF, scenario_conditions
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 999
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
conf = {"model": dict(CONF["model"])}; conf["model"]["diffusion"] = {"num_steps": steps, "beta_schedule": "cosine"}
torch.manual_seed(0)
m = BiologyAwareDiffusionModel(62, 5054, 26, 3, conf).cuda().eval()
m.sampler, m.chain_variant = "chain", "squad"
eng = m._engine()
fn = L.lib().osd_dbg_chain_stamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p]
buf = torch.zeros(4096 * 4 * 16, dtype=torch.int64, device="cuda")
cond = scenario_conditions(n, 0).cuda()
m.sample(cond, n, seed=1)
L.check(fn(eng.handle, L.ptr(buf)))
torch.cuda.synchronize(); t0 = time.perf_counter()
m.sample(cond, n, seed=2)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
assert m.last_chain_variant == "squad"
s = buf.cpu().numpy().reshape(-1, 4, 16).astype(float)
s = s[s[:, 0, 12] > 0]
print(f"n={n} T={steps}: {dt*1e3:.1f} ms = {dt/steps*1e6:.1f} us/step, {len(s)} workgroups, clock ~{s[:, 0, 11].mean()/dt/1e9:.2f} GHz")
names = ["input_proj K loop + slab stores", "sync 1", "reduce", "sync 2", "layers: K loops", "layers: partials -> LDS, GroupNorm + SiLU (wave 0)",
         "layers: syncs", "output_proj: operand -> LDS", "output_proj: K loops", "posterior epilogues", "end-of-step barrier + prime"]
clk = s[:, 0, 11].mean() / dt / 1e6      # cycles per us
print(f"{'phase':58s}" + "".join(f"   wave {w} us/step" for w in range(4)))
for i, nm in enumerate(names):
    print(f"{nm:58s}" + "".join(f"{s[:, w, i].mean() / steps / clk:17.2f}" for w in range(4)))
print(f"{'total':58s}" + "".join(f"{s[:, w, 11].mean() / steps / clk:17.2f}" for w in range(4)))
for g in range(8):
    sel = s[g::8]
    print(f"  workgroup g={g}: " + " ".join(f"{sel[:, 0, i].mean() / steps / clk:6.2f}" for i in range(11)))
