#!/usr/bin/env python3
"""Inner-loop micro-benchmark (diagnostic): TFLOP/s of the 128x128 tile loop with the staging
pieces switched on one at a time."""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import torch
from osteosarcoma_diffusionmodel_amd import _lib as L
from helpers import RawHandle

rh = RawHandle()
lib = L.lib()
fn = lib.osd_dbg_mfma_rate
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
src = torch.randn(65536 + 1024 * 128 * 2000 + 4096, device="cuda")
dst = torch.zeros(1024 * 128 * 512, device="cuda")
names = {0: "MFMA only", 1: "+LDS fragment reads", 2: "+barrier/step", 3: "+glds DMA (L2-hot)", 4: "+glds DMA (x rows from HBM)", 5: "+DMA +quad stores [p][f]", 6: "+DMA +row stores"}
for grid in (512, 1024):
    for mode in (3, 4):
        for nk in (8, 64):
            ms = C.c_float()
            L.check(fn(rh.h, mode, nk, grid, L.ptr(src), L.ptr(dst), C.byref(ms)))
            flops = grid * 4 * 64 * nk * 4096.0       # waves * MFMAs/step * steps * flop/MFMA
            print(f"grid={grid:5d} nk={nk:3d} {names[mode]:22s} {ms.value*1e3:9.1f} us  {flops/ms.value/1e9:7.1f} TFLOP/s")

fn16 = lib.osd_dbg_mfma_rate16
fn16.restype = C.c_int
fn16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
for grid in (512, 1024):
    for stages in (2, 3):
        for nk in (16, 128):
            ms = C.c_float()
            L.check(fn16(rh.h, stages, nk, grid, L.ptr(src), L.ptr(dst), C.byref(ms)))
            flops = grid * 4 * 32 * nk * 4096.0
            print(f"grid={grid:5d} nk16={nk:3d} BK=16 {stages}-stage DMA (x rows from HBM)  {ms.value*1e3:9.1f} us  {flops/ms.value/1e9:7.1f} TFLOP/s")
