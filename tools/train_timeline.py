#!/usr/bin/env python3
"""Diagnostic: one training step's kernel timeline from a rocprofv3 --kernel-trace CSV (start/end per kernel, gaps, streams)."""
import csv, glob, sys, re
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at each k_mixup triple's first kernel; take the last complete step
starts = [i for i, r in enumerate(rows) if "k_mixup" in r["Kernel_Name"]]
firsts = [s for j, s in enumerate(starts) if j == 0 or s - starts[j - 1] > 3]
a, b = firsts[-2], firsts[-1]
t0 = int(rows[a]["Start_Timestamp"])
def short(n):
    n = re.sub(r"\(.*", "", n.replace("void ", "").replace("osd::", ""))
    return n[:70]
print(f"step: {b - a} kernels, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
prev_end = {}
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = r.get("Stream_Id", r.get("Queue_Id", "?"))
    gap = s - prev_end.get(q, s)
    prev_end[q] = e
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f} gap {gap/1e3:6.1f} q{q} {short(r['Kernel_Name'])}")
