#!/usr/bin/env python3
"""Per-kernel-class table of the PMC passes of tools/train_pmc.sh (training step, BASELINE config 2).
Usage: train_pmc_summary.py <dir> <round tag>
Medians over the launches of a class (the first launches of a process are cold: their GRBM_GUI_ACTIVE is many times the
steady-state value, which wrecks a mean); ratios are formed per dispatch, inside one pass, before the median."""
import collections, csv, glob, re, statistics, sys
root, tag = sys.argv[1], sys.argv[2]
disp = collections.defaultdict(lambda: collections.defaultdict(dict))     # pass file -> (kernel, dispatch) -> counter -> value


def klass(name):
    name = re.sub(r"\(.*$", "", name).replace("osd::", "").replace("void ", "")
    name = name.replace("Tile<64, 64, 32, 32>", "T64").replace("Tile<64, 128, 64, 32>", "T64x128").replace("Tile<128, 128, 64, 64>", "T128")
    return name.strip()[:90]


for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = disp[f][(klass(r["Kernel_Name"]), r["Dispatch_Id"])]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["_dur"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])

per = collections.defaultdict(lambda: collections.defaultdict(list))      # kernel -> metric -> per-dispatch values
for f, dd in disp.items():
    for (k, _), c in dd.items():
        if "GRBM_GUI_ACTIVE" in c:
            per[k]["us"].append(c["_dur"] / 1e3)
            simd_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 1024
            if simd_cycles > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                per[k]["mfma_busy"].append(c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles)
                per[k]["mfma_us"].append(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / 2400.0)
            if c.get("SQ_WAVE_CYCLES"):
                per[k]["wait"].append(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"])
        for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_LDS_BANK_CONFLICT", "FETCH_SIZE", "WRITE_SIZE"):
            if name in c:
                per[k][name].append(c[name])
        if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0) > 0:
            per[k]["l2hit"].append(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]))


def med(k, m):
    v = per[k].get(m)
    return statistics.median(v) if v else float("nan")


print(f"# {tag} -- PMC counters of the training step's kernels (B = 4096, D = 2000; `tools/train_pmc.sh`)\n")
print("Per launch, MEDIAN over the launches of each kernel class (`rocprofv3 --kernel-trace --pmc`, one counter set per pass; under --pmc")
print("kernels run serialised and a few us slower than in the timeline).  `mfma_busy` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024")
print("SIMDs), formed per dispatch; `mfma us` = the same cycles / 1024 SIMDs / 2.4 GHz (time the matrix pipes would need alone);")
print("traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of MI355X_MICROARCH.md); `wait` = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.\n")
print("| kernel | launches | us | mfma us | mfma_busy | wait | VALU insts | SALU insts | LDS insts | VMEM insts | LDS conflicts | FETCH KiB | WRITE KiB | traffic MB | L2 hit |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
rows = []
for k in per:
    n = len(per[k].get("us", []))
    if n == 0:
        continue
    fetch, write = med(k, "FETCH_SIZE"), med(k, "WRITE_SIZE")
    rows.append((med(k, "us") * n, f"| {k} | {n} | {med(k, 'us'):.1f} | {med(k, 'mfma_us'):.1f} | {med(k, 'mfma_busy'):.3f} | {med(k, 'wait'):.3f} | {med(k, 'SQ_INSTS_VALU'):.0f} | "
                 f"{med(k, 'SQ_INSTS_SALU'):.0f} | {med(k, 'SQ_INSTS_LDS'):.0f} | {med(k, 'SQ_INSTS_VMEM'):.0f} | {med(k, 'SQ_LDS_BANK_CONFLICT'):.0f} | "
                 f"{fetch:.0f} | {write:.0f} | {(2 * fetch + write) * 1024 / 1e6:.1f} | {med(k, 'l2hit'):.3f} |"))
for _, line in sorted(rows, reverse=True):
    print(line)
