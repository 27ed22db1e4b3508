#!/usr/bin/env python3
"""Diagnostic: one training step's kernel timeline from a rocprofv3 --kernel-trace CSV (start/end per kernel, gaps, streams)."""
import csv, glob, sys, re
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at the kernel that follows an optimizer kernel (k_adamw); take a steady-state step of the path named by argv[2]
# ("src": rows from the resident dataset, the default; "mixup": the host hand-over path with its own mixup kernel) -- the step
# that sits in the middle of the matching ones
want = sys.argv[2] if len(sys.argv) > 2 else "src"
ends = [i for i, r in enumerate(rows) if "k_adamw" in r["Kernel_Name"]]
steps = [(ends[j] + 1, ends[j + 1] + 1) for j in range(len(ends) - 1)]
def kind(a, b):
    names = " ".join(r["Kernel_Name"] for r in rows[a:b])
    return "src" if "k_q_sample_src" in names else ("mixup" if "k_mixup" in names else "other")
match = [(a, b) for a, b in steps if kind(a, b) == want and b - a < 80]
a, b = match[len(match) // 2]
while "elementwise" in rows[a]["Kernel_Name"] or "index" in rows[a]["Kernel_Name"]:      # torch's loss accumulation / index gather belong to the previous step's tail / this step's head
    a += 1
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
    wgs = int(r.get("Grid_Size_X", 0) or 0) // max(1, int(r.get("Workgroup_Size_X", 1) or 1)) * max(1, int(r.get("Grid_Size_Y", 1) or 1))
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f} gap {gap/1e3:6.1f} q{q} wg{wgs:<5d} {short(r['Kernel_Name'])}")
