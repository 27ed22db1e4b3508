#!/usr/bin/env python3
"""Aggregate the counter_collection.csv files of tools/round_pmc.sh into the per-kernel table of profiles/rNN_pmc.md and
the traffic json bench.py reads (profiles/rNN_traffic.json).  Usage: pmc_summary.py <dir> <round tag> <chain patient-steps per launch>"""
import csv, glob, json, collections, sys
root, tag, chain_units = sys.argv[1], sys.argv[2], float(sys.argv[3])
vals = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel class -> counter -> per-dispatch values
dur = collections.defaultdict(list)


def klass(name):
    if "squad_chain_kernel" in name: return "squad_chain_kernel"
    if "panel_chain_kernel" in name: return "panel_chain_kernel"
    if "chain_kernel" in name: return "chain_kernel"
    if "wgrad_group_kernel" in name: return "wgrad_group_kernel"
    if "gemm_bf3_kernel" in name:
        if "EpiB3Gn<32" in name: return "bf16x3 GnSilu<32>"
        if "EpiB3Gn<64" in name: return "bf16x3 GnSilu<64>"
        if "EpiB3Input" in name: return "bf16x3 Input"
        if "EpiB3Post" in name: return "bf16x3 Posterior"
        return None
    if "gemm_glds_kernel" not in name or "128, 128, 64, 64" not in name: return None
    if "EpiGnSilu<32" in name: return "GnSilu<32> glds"
    if "EpiGnSilu<64" in name: return "GnSilu<64> glds"
    if "EpiInput" in name: return "Input glds"
    if "EpiPosterior" in name: return "Posterior glds"
    return None


for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = klass(r["Kernel_Name"])
        if k is None: continue
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))


def avg(k, c):
    v = vals[k].get(c)
    return sum(v) / len(v) if v else float("nan")


print("| kernel | launches | avg ms (profiled) | mfma_busy | LDS bank conflicts | FETCH_SIZE KiB | WRITE_SIZE KiB | traffic MB/launch | L2 hit | VALU insts |")
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
    print(f"| {k} | {n} | {sum(dur[k]) / max(len(dur[k]), 1) / 1e6:.3f} | {busy:.3f} | {avg(k, 'SQ_LDS_BANK_CONFLICT'):.0f} | {fetch:.0f} | {write:.0f} | "
          f"{tr / 1e6:.1f} | {hit / (hit + miss):.3f} | {avg(k, 'SQ_INSTS_VALU'):.0f} |")
units = {k: (chain_units if k in ("chain_kernel", "panel_chain_kernel") else (2976.0 * 100 if k == "squad_chain_kernel" else 32768.0)) for k in traffic}      # squad pass: chain_run.py 2976 100
# the library these counters were collected on: bench.py only quotes a traffic figure for the build it is running (sha256 of the .so)
import hashlib, os
from pathlib import Path
lib_path = Path(os.environ.get("OSDIFF_LIB", Path(__file__).resolve().parent.parent / "osteosarcoma_diffusionmodel_amd" / "lib" / "libosdiff.so"))
lib_sha = hashlib.sha256(lib_path.read_bytes()).hexdigest() if lib_path.exists() else None
tj = {"rows_per_launch": 32768, "units_per_launch": units, "traffic_bytes_per_launch": traffic,
      "unit_of": {"chain_kernel": "patient-steps", "default": "rows"}, "source": f"profiles/{tag}_pmc.md", "library_sha256": lib_sha}
print()
print("traffic_json:", json.dumps(tj))
Path(root, f"{tag}_traffic.json").write_text(json.dumps(tj, indent=1))
