#!/usr/bin/env python3
"""Summarise the per-workgroup cycle counters of the persistent backward kernel (OSD_BWD_STAMPS=<file>)."""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1])
a = a[a[:, 3] > 0]
clk = 100e6 if len(sys.argv) < 3 else float(sys.argv[2])       # s_memtime ticks at 100 MHz on gfx950
us = lambda c: c / clk * 1e6
print(f"workgroups {len(a)}; total (max) {us(a[:, 3].max()):.1f} us, mean {us(a[:, 3].mean()):.1f}")
print(f"per workgroup mean: scheduler {us(a[:, 0].mean()):.1f} us, dgrad {us(a[:, 1].mean()):.1f} us in {a[:, 4].mean():.1f} units, wgrad {us(a[:, 2].mean()):.1f} us in {a[:, 5].mean():.1f} items")
print(f"per unit: dgrad {us(a[:, 1].sum() / max(a[:, 4].sum(), 1)):.2f} us, wgrad {us(a[:, 2].sum() / max(a[:, 5].sum(), 1)):.2f} us, scheduler per unit {us(a[:, 0].sum() / max(a[:, 4].sum() + a[:, 5].sum(), 1)):.2f} us")
print(f"units total: dgrad {int(a[:, 4].sum())}, wgrad {int(a[:, 5].sum())}")
print(f"publish (drain + barrier + release + counters) per unit: {us(a[:, 7].sum() / max(a[:, 4].sum() + a[:, 5].sum(), 1)):.2f} us")
names = ["GN32", "GN32+drop", "GN64", "GN64+drop", "plain"]
for i, nm in enumerate(names):
    n = a[:, 13 + i].sum()
    if n:
        print(f"  dgrad {nm}: {int(n)} units, {us(a[:, 8 + i].sum() / n):.2f} us each")
sys.exit(0)
for x in range(8):
    m = a[:, 6] == x
    if m.any():
        print(f"  xcd {x}: {m.sum()} wgs, dgrad units {int(a[m, 4].sum())}, wgrad items {int(a[m, 5].sum())}, sched {us(a[m, 0].mean()):.1f} us")
