#!/usr/bin/env python3
"""Diagnostic (needs `make DIAG=1`): where the cycles of the LDS-resident chain kernel's workgroups go (wave 0's clock)."""
import ctypes as C, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import CONF, scenario_conditions
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
conf = {"model": dict(CONF["model"])}; conf["model"]["diffusion"] = {"num_steps": steps, "beta_schedule": "cosine"}
torch.manual_seed(0)
m = BiologyAwareDiffusionModel(50, 1900, 50, 3, conf).cuda().eval()
m.sampler, m.chain_variant = "chain", "panel"
eng = m._engine()
fn = L.lib().osd_dbg_chain_stamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p]
buf = torch.zeros(1024 * 64, dtype=torch.int64, device="cuda")
cond = scenario_conditions(n, 0).cuda()
m.sample(cond, n, seed=1)
L.check(fn(eng.handle, L.ptr(buf)))
torch.cuda.synchronize(); t0 = time.perf_counter()
m.sample(cond, n, seed=2)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
assert m.last_chain_variant == "panel"
s = buf.cpu().numpy().reshape(-1, 64); s = s[s[:, 5] > 0].astype(float)
tot, units = s[:, 4], s[:, 5]
print(f"n={n} T={steps}: {dt*1e3:.1f} ms, {n*steps/dt/1e6:.2f} M patient-steps/s, {len(s)} workgroups, units/wg {units.min():.0f}..{units.max():.0f}, "
      f"cycles/unit {tot.sum()/units.sum():.0f}, clock ~{tot.mean()/dt/1e9:.2f} GHz")
# ideal MFMA cycles per unit: 64 rows x flop/row / (256 flop/cycle/CU)
names = [("dependency wait + queue", 0), ("input_proj (all)", 1), ("  of which epilogue", 8), ("GroupNorm layers (all)", 2),
         ("  256-wide K loops", 9), ("  256-wide epilogues", 10), ("  512-wide K loops", 11), ("  512-wide epilogues", 12), ("  spill reload", 15),
         ("output_proj + posterior (all)", 3), ("  K loops", 13), ("  posterior epilogues", 14)]
for name, col in names:
    print(f"  {name:32s} {100*s[:, col].sum()/tot.sum():6.2f} %   {s[:, col].sum()/units.sum():10.0f} cycles/unit")
print("  layer 8 (512 -> 512, ideal 131072 MFMA cycles), per wave: setup | K loop | DMA wait | barrier | epilogue | barrier")
for w in range(4):
    c = s[:, 16 + 8 * w: 22 + 8 * w].sum(axis=0) / units.sum()
    print(f"    wave {w}: " + "".join(f"{v:10.0f}" for v in c))
print("  input_proj chunk 3 (256 k; ideal 32768 MFMA cycles per SIMD), per wave: K loop | DMA wait | barrier")
for w in range(4):
    c = [s[:, 16 + 8 * w + 6].sum() / units.sum(), s[:, 16 + 8 * w + 7].sum() / units.sum(), s[:, 48 + w].sum() / units.sum()]
    print(f"    wave {w}: " + "".join(f"{v:10.0f}" for v in c))
mf = {"input": 64*2*2048*256/256, "gn256": 64*2*(512*256+256*256*4+1024*256)/256 if False else 0}
