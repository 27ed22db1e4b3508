#!/usr/bin/env python3
"""Diagnostic (needs `make DIAG=1`, OSDIFF_LIB=<that library>): where the cycles of the training forward's squad kernel go, at one
and at two workgroups per CU.   train_squad_stamps.py [rows ...]"""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import CONF
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, _lib as L
rows = [int(v) for v in sys.argv[1:]] or [2048, 4096]
torch.manual_seed(0)
m = BiologyAwareDiffusionModel(50, 1900, 50, 3, CONF).cuda().train()
eng = m._engine()
fn = L.lib().osd_dbg_chain_stamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p]
names = ["K loops", "partials -> LDS + barrier", "epilogue (+ unit stores)", "arrive .. released (row-major stores, next weights, poll)", "h0 -> units"]
for n in rows:
    x = torch.randn(n, 2000, device="cuda"); c = torch.randn(n, 3, device="cuda")
    buf = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        m(x, c, seed=1)
        L.check(fn(eng.handle, L.ptr(buf)))
        m(x, c, seed=2)
    torch.cuda.synchronize()
    s = buf.cpu().numpy().reshape(-1, 4, 8).astype(float)
    s = s[s[:, 0, 5] > 0]
    clk = 2.1e3          # cycles per us (the part holds ~2.0-2.2 GHz under fp32 MFMA load)
    print(f"rows {n}: {len(s)} workgroups; us per launch at {clk/1e3:.1f} GHz, mean over workgroups, per wave")
    for i, nm in enumerate(names):
        print(f"  {nm:58s}" + "".join(f"{s[:, w, i].mean() / clk:9.1f}" for w in range(4)))
    print(f"  {'whole kernel':58s}" + "".join(f"{s[:, w, 5].mean() / clk:9.1f}" for w in range(4)))
    L.check(fn(eng.handle, None))
