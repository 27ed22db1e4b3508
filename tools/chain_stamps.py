#!/usr/bin/env python3
"""Diagnostic (needs `make DIAG=1`): cycle shares of the persistent chain kernel's workgroups at the config-3 shape."""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import CONF, scenario_conditions
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
stagger = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
conf = {"model": dict(CONF["model"])}; conf["model"]["diffusion"] = {"num_steps": steps, "beta_schedule": "cosine"}
torch.manual_seed(0)
m = BiologyAwareDiffusionModel(50, 1900, 50, 3, conf).cuda().eval()
m.sampler, m.chain_stagger = "chain", stagger
if len(sys.argv) > 4: m.chain_grid = int(sys.argv[4])
eng = m._engine()
fn = L.lib().osd_dbg_chain_stamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p]
buf = torch.zeros(1024 * 64, dtype=torch.int64, device="cuda")
cond = scenario_conditions(n, 0).cuda()
m.sample(cond, n, seed=1)                       # warm-up, unstamped
L.check(fn(eng.handle, L.ptr(buf)))
torch.cuda.synchronize(); import time; t0 = time.perf_counter()
m.sample(cond, n, seed=2)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
s = buf.cpu().numpy().reshape(-1, 64); s = s[s[:, 5] > 0]
tot = s[:, 4].astype(float)
print(f"n={n} T={steps} stagger={stagger}: {dt*1e3:.1f} ms, {n/dt*steps/1000:.0f} patient-ksteps/s, {len(s)} workgroups, units/wg {s[:,5].min()}..{s[:,5].max()}")
for name, col in (("dependency wait", 0), ("tile prologue (first DMA stage)", 1), ("K loop", 2), ("epilogue + drain", 3)):
    sh = s[:, col] / tot
    print(f"  {name:34s} {100*sh.mean():6.2f} %  (min {100*sh.min():.2f}, max {100*sh.max():.2f})   {s[:, col].mean()/s[:,5].mean():12.0f} cycles/unit")
print(f"  kernel cycles / unit {tot.mean()/s[:,5].mean():.0f}; clock ~ {tot.mean()/dt/1e9:.2f} GHz")

# per layer kind: cycles per tile in each phase (workgroup mean)
print("  per tile, by epilogue kind:   K loop | epilogue to last store issued | store drain | barrier | boundary first stage | tiles/unit")
for kind, name in enumerate(("input_proj", "GroupNorm 32", "GroupNorm 64", "posterior")):
    c = s[:, 8 + 10 * kind: 18 + 10 * kind].astype(float).sum(axis=0)
    if c[5] == 0:
        continue
    print(f"    {name:14s}" + "".join(f"{c[i] / c[5]:10.0f}" for i in range(5)) + f"   {c[5] / s[:, 5].sum():6.1f}")
    if c[6]:
        print("        GroupNorm epilogue: first-stage A issue + bias loads landed | statistics | gamma/beta landed | normalise + SiLU + stores issued:"
              + "".join(f"{c[i] / c[5]:9.0f}" for i in range(6, 10)))

# placement: units completed per CU (both workgroups) and per XCD
import collections
hw, xcc = s[:, 6], s[:, 7]
cu_key = [(int(x), int((h >> 8) & 0xFF)) for h, x in zip(hw, xcc)]
per_cu = collections.defaultdict(list)
for k, u in zip(cu_key, s[:, 5]):
    per_cu[k].append(int(u))
sizes = collections.Counter(len(v) for v in per_cu.values())
tot = np.array([sum(v) for v in per_cu.values()])
print(f"  CUs used {len(per_cu)}; workgroups per CU {dict(sizes)}; units per CU min {tot.min()} mean {tot.mean():.1f} max {tot.max()}")
split = [abs(v[0] - v[1]) for v in per_cu.values() if len(v) == 2]
print(f"  |difference| between the two workgroups of a CU: mean {np.mean(split):.1f} max {np.max(split)}")
per_x = collections.defaultdict(int)
for (x, _), v in per_cu.items():
    per_x[x] += sum(v)
print("  units per XCD:", dict(sorted(per_x.items())))
