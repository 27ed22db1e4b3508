#!/usr/bin/env python3
"""Diagnostic: hardware placement (XCC, SE, CU, SIMD, wave slot) of the workgroups of a 2-per-CU launch."""
import ctypes as C, sys, collections
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import torch
from osteosarcoma_diffusionmodel_amd import _lib as L
from helpers import RawHandle
rh = RawHandle(); lib = L.lib()
fn = lib.osd_dbg_census; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
out = torch.zeros(grid * 8, dtype=torch.int32, device="cuda")
L.check(fn(rh.h, grid, L.ptr(out)))
o = out.cpu().numpy().astype("uint32").reshape(grid, 4, 2)
cus = collections.defaultdict(list)
for b in range(grid):
    for w in range(4):
        hw, xcc = int(o[b, w, 0]), int(o[b, w, 1]) & 0xF
        wave, simd, pipe, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 6) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        cus[(xcc, se, sh, cu)].append((b, w, simd, wave))
print("distinct CUs:", len(cus))
for i, (k, v) in enumerate(sorted(cus.items())):
    if i < 6: print(k, sorted(v))
pairs = collections.Counter()
for k, v in cus.items():
    blocks = sorted(set(b for b, *_ in v))
    slots = {b: sorted(set(wave for bb, w, simd, wave in v if bb == b)) for b in blocks}
    pairs[str([slots[b] for b in blocks])] += 1
    if len(blocks) == 2: pairs["delta_block=%d" % (blocks[1] - blocks[0])] += 1
print(pairs.most_common(12))
