#!/usr/bin/env python3
"""Per-kernel-class medians of the counters tools/split_pmc.sh collected on the bf16x3 engine (profiles/rNN_split_pmc.md)."""
import collections, csv, glob, statistics, sys
root = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)


def klass(name):
    if "gemm_bf3_kernel" not in name: return None
    for k, v in (("EpiB3Gn<32", "Linear+GN+SiLU 256-wide"), ("EpiB3Gn<64", "Linear+GN+SiLU 512-wide"), ("EpiB3Input", "input_proj"), ("EpiB3Post", "output_proj+posterior")):
        if k in name: return v
    return None


for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = klass(r["Kernel_Name"])
        if k is None: continue
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))


def med(k, c):
    v = vals[k].get(c)
    return statistics.median(v) if v else float("nan")


print("bf16x3 engine, 65 536 rows, one reverse step per row of launches; medians over launches.  `mfma_busy` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs / 8 XCD-summed);")
print("`wait` = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls), `parked` = SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt / barrier); traffic = 2 x FETCH_SIZE + WRITE_SIZE.\n")
print("| kernel | launches | us | mfma_busy | coexec / mfma_busy | wait | parked | VALU insts | MFMA insts | LDS insts | VMEM insts | SALU insts | LDS conflicts | traffic MB | L2 hit |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for k in sorted(vals):
    gui = med(k, "GRBM_GUI_ACTIVE")
    busy = med(k, "SQ_VALU_MFMA_BUSY_CYCLES") / (gui / 8 * 1024)
    co = med(k, "SQ_VALU_MFMA_COEXEC_CYCLES") / max(med(k, "SQ_VALU_MFMA_BUSY_CYCLES"), 1)
    wc = med(k, "SQ_WAVE_CYCLES")
    tr = (2 * med(k, "FETCH_SIZE") + med(k, "WRITE_SIZE")) * 1024
    hit, miss = med(k, "TCC_HIT_sum"), med(k, "TCC_MISS_sum")
    print(f"| {k} | {len(vals[k].get('GRBM_GUI_ACTIVE', []))} | {statistics.median(dur[k]) / 1e3 if dur[k] else float('nan'):.1f} | {busy:.3f} | {co:.3f} | "
          f"{med(k, 'SQ_WAIT_INST_ANY') / wc:.3f} | {med(k, 'SQ_WAIT_ANY') / wc:.3f} | {med(k, 'SQ_INSTS_VALU'):.3g} | {med(k, 'SQ_INSTS_MFMA'):.3g} | "
          f"{med(k, 'SQ_INSTS_LDS'):.3g} | {med(k, 'SQ_INSTS_VMEM'):.3g} | {med(k, 'SQ_INSTS_SALU'):.3g} | {med(k, 'SQ_LDS_BANK_CONFLICT'):.3g} | {tr / 1e6:.1f} | {hit / (hit + miss):.3f} |")
