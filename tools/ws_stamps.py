#!/usr/bin/env python3
"""Diagnostic: cycles per tile of the MFMA waves and per epilogue of the shadow waves in gemm_ws_kernel."""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np, torch
from osteosarcoma_diffusionmodel_amd import _lib as L
from helpers import RawHandle
rh = RawHandle(); lib = L.lib()
L.check(lib.osd_set_option(rh.h, b"wave_specialized", 2))
fn = lib.osd_dbg_stamp_gn; fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
for (K, N, n) in ((256, 256, 65536), (512, 512, 32768), (512, 512, 65536)):
    x = torch.randn(n, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda"); ga = torch.ones(N, device="cuda"); be = torch.zeros(N, device="cuda")
    y = torch.empty(n, N, device="cuda")
    st = torch.zeros(256 * 16, dtype=torch.int64, device="cuda")
    for _ in range(3):
        L.check(fn(rh.h, L.ptr(x), K, L.ptr(w), L.ptr(b), L.ptr(ga), L.ptr(be), n, N, L.ptr(y), L.ptr(st)))
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(256, 4, 4).astype(np.float64)
    m, e = s[:, 0, :], s[:, 1, :]
    tiles = (n // 128) * (N // 128) / 256
    print(f"K={K} N={N} rows={n} tiles/WG={tiles:.1f}: MFMA wave: tile0 {np.mean(m[:,1]-m[:,0]):8.0f}  tile1 {np.mean(m[:,2]-m[:,1]):8.0f}  total {np.mean(m[:,3]-m[:,0]):9.0f}"
          f" | epilogue wave: finish {np.mean(e[:,1]-e[:,0]):8.0f}  drain {np.mean(e[:,2]-e[:,1]):8.0f} cycles (100 MHz ticks x21 if memtime is the 100 MHz counter)")
