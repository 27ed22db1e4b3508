#!/usr/bin/env python3
"""Aggregate the counter_collection.csv files of tools/round_pmc.sh into the per-kernel table of profiles/r01_pmc.md."""
import csv, glob, re, sys, json, collections
root = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel class -> counter -> per-dispatch values
dur = collections.defaultdict(list)
def klass(name):
    if "gemm_glds_kernel" not in name or "128, 128, 64, 64" not in name: return None
    m = re.search(r"Epi(\w+?)(?:<(\d+), (?:true|false)>)?\s*>?\(", name.replace("osd::", ""))
    if "EpiGnSilu<32" in name: return "GnSilu<32> glds"
    if "EpiGnSilu<64" in name: return "GnSilu<64> glds"
    if "EpiInput" in name: return "Input glds"
    if "EpiPosterior" in name: return "Posterior glds"
    return None
for f in glob.glob(root + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = klass(r["Kernel_Name"])
        if k is None: continue
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE",):
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
def avg(k, c):
    v = vals[k].get(c); return sum(v) / len(v) if v else float("nan")
print("| kernel (Tile 128x128, LDS-DMA) | launches | avg ns (profiled) | mfma_busy | LDS bank conflicts | FETCH_SIZE KiB | WRITE_SIZE KiB | traffic MB/launch | L2 hit | VALU insts |")
print("|---|---|---|---|---|---|---|---|---|---|")
traffic = {}
for k in sorted(vals):
    gui = avg(k, "GRBM_GUI_ACTIVE")
    busy = avg(k, "SQ_VALU_MFMA_BUSY_CYCLES") / (gui / 8 * 1024)
    fetch, write = avg(k, "FETCH_SIZE"), avg(k, "WRITE_SIZE")
    tr = (2 * fetch + write) * 1024
    traffic[k] = tr
    hit, miss = avg(k, "TCC_HIT_sum"), avg(k, "TCC_MISS_sum")
    n = len(vals[k].get("GRBM_GUI_ACTIVE", []))
    print(f"| {k} | {n} | {sum(dur[k]) / max(len(dur[k]), 1):.0f} | {busy:.3f} | {avg(k, 'SQ_LDS_BANK_CONFLICT'):.0f} | {fetch:.0f} | {write:.0f} | {tr / 1e6:.1f} | {hit / (hit + miss):.3f} | {avg(k, 'SQ_INSTS_VALU'):.0f} |")
print()
print("traffic_json:", json.dumps({"rows_per_launch": 32768, "traffic_bytes_per_launch": traffic, "source": "profiles/r01_pmc.md"}))
