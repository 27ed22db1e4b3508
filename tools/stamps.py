#!/usr/bin/env python3
"""Diagnostic: where a Linear+GroupNorm+SiLU workgroup spends its cycles (s_memtime stamps per wave)."""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np, torch
from osteosarcoma_diffusionmodel_amd import _lib as L
from helpers import RawHandle
rh = RawHandle(); lib = L.lib()
fn = lib.osd_dbg_stamp_gn; fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
shapes = ((256, 256, 32768), (512, 512, 32768), (512, 512, 65536), (256, 256, 131072))
if len(sys.argv) > 1 and sys.argv[1] == "train":      # the training batch (BASELINE config 2): one workgroup per CU
    shapes = ((256, 256, 4096), (512, 256, 4096), (256, 512, 4096), (512, 512, 4096), (1024, 256, 4096))
for (K, N, n) in shapes:
    x = torch.randn(n, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda"); ga = torch.ones(N, device="cuda"); be = torch.zeros(N, device="cuda")
    y = torch.empty(n, N, device="cuda")
    grid = 8192                                      # upper bound over the tile shapes the launcher may pick; unused entries stay 0
    st = torch.zeros(grid * 16, dtype=torch.int64, device="cuda")
    for _ in range(3):
        st.zero_()
        L.check(fn(rh.h, L.ptr(x), K, L.ptr(w), L.ptr(b), L.ptr(ga), L.ptr(be), n, N, L.ptr(y), L.ptr(st)))
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(grid, 4, 4).astype(np.float64)
    s = s[s[:, 0, 0] > 0]
    pro, loop, epi = s[:, :, 1] - s[:, :, 0], s[:, :, 2] - s[:, :, 1], s[:, :, 3] - s[:, :, 2]
    t0 = s[:, :, 0].min(); tend = s[:, :, 3].max()
    print(f"K={K} N={N} rows={n} WGs={len(s)}: prologue {pro.mean():8.0f}  K-loop {loop.mean():8.0f} ({loop.mean()/(K/32):6.0f}/step)  epilogue+drain {epi.mean():8.0f} cycles;"
          f" kernel span {tend - t0:9.0f} cycles; start spread {(s[:, :, 0].max() - t0):8.0f}")
