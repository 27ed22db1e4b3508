#!/usr/bin/env python3
"""Register / scratch audit of every kernel in csrc/*.hip (no GPU needed): compiles each translation unit for gfx950 with
-Rpass-analysis=kernel-resource-usage and tabulates VGPRs, SGPRs, spills and scratch per kernel.

    python tools/isa_check.py [--md profiles/r02_isa_resources.md] [--strict]

--strict exits 1 when a hot-path kernel (gemm_glds_kernel<...>, chain_kernel) spills a VGPR."""
import argparse
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "osteosarcoma_diffusionmodel_amd" / "csrc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I.", "--cuda-device-only",
         "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]
KEYS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ",
        "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def audit(src):
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS[:-1] + ["/dev/null", str(src.name)], cwd=CSRC, capture_output=True, text=True)
    kernels, cur = [], None
    for ln in r.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", ln)
        if m:
            cur = {"mangled": m.group(1), "file": src.name}
            kernels.append(cur)
            continue
        m = re.search(r"remark:\s+(.+?): (\d+)", ln)
        if m and cur is not None and m.group(1) in KEYS:
            cur[KEYS[m.group(1)]] = int(m.group(2))
    if r.returncode != 0:
        print(r.stderr[-2000:], file=sys.stderr)
        raise SystemExit(f"{src.name}: compile failed")
    return kernels


def lane_moves_in_mfma_blocks(src):
    """SGPR spills are v_writelane / v_readlane moves; they only cost matrix-pipe time where they sit between MFMAs (an fp32
    MFMA shares its SIMD's issue with VALU work).  Per kernel: (lane moves in basic blocks that contain a v_mfma, all lane moves)."""
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I.",
                        "--cuda-device-only", "-S", "-o", "-", str(src.name)], cwd=CSRC, capture_output=True, text=True)
    out, cur, blocks = {}, None, []
    for ln in r.stdout.splitlines():
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur, blocks = m.group(1), [[]]
            continue
        if cur is None:
            continue
        if ln.startswith(".LBB") or ln.startswith(".Lfunc_end"):
            blocks.append([])
        blocks[-1].append(ln)
        if ln.startswith(".Lfunc_end"):
            hot = sum(sum(1 for x in b if "v_writelane" in x or "v_readlane" in x) for b in blocks if any("v_mfma" in x for x in b))
            tot = sum(1 for b in blocks for x in b if "v_writelane" in x or "v_readlane" in x)
            out[cur] = (hot, tot)
            cur = None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--md", default="")
    ap.add_argument("--strict", action="store_true")
    args = ap.parse_args()
    srcs = sorted(CSRC.glob("*.hip"))
    with ThreadPoolExecutor(4) as ex:
        ks = [k for res in ex.map(audit, srcs) for k in res]
        moves = {}
        for res in ex.map(lane_moves_in_mfma_blocks, srcs):
            moves.update(res)
    for k in ks:
        k["lane_hot"], k["lane_all"] = moves.get(k["mangled"], (0, 0))
    for k, name in zip(ks, demangle([k["mangled"] for k in ks])):
        k["name"] = re.sub(r"\(.*\)$", "", name.replace("void ", "").replace("osd::", ""))
    hot = [k for k in ks if "gemm_glds_kernel" in k["name"] or "chain_kernel" in k["name"]]
    bad = [k for k in hot if k.get("vgpr_spill", 0)]
    scratchy = [k for k in hot if k.get("scratch", 0) and not k.get("vgpr_spill", 0)]
    lines = ["# Kernel register audit (gfx950, hipcc -O3, `tools/isa_check.py`)", "",
             f"{len(ks)} kernels in {len(srcs)} translation units; hot-path kernels (gemm_glds_kernel<...>, chain_kernel): {len(hot)}, "
             f"of which {len(bad)} spill a VGPR; {len(scratchy)} use a few bytes of scratch without spilling (a small local array).", "",
             "`lane moves` = v_writelane / v_readlane instructions (SGPR spill traffic): `in MFMA blocks` counts those in basic blocks that "
             "also hold a v_mfma -- only there do they take issue slots from the matrix pipe -- against all of them in the kernel.", "",
             f"Hot-path kernels with lane moves inside an MFMA block: {sum(1 for k in hot if k['lane_hot'])}.", "",
             "| kernel | file | VGPR | SGPR | VGPR spill | SGPR spill | lane moves in MFMA blocks / all | scratch B/lane | waves/SIMD |", "|---|---|---|---|---|---|---|---|---|"]
    for k in sorted(ks, key=lambda k: (-(k.get("vgpr_spill", 0)), -k.get("vgpr", 0), k["name"])):
        lines.append(f"| `{k['name']}` | {k['file']} | {k.get('vgpr', '')} | {k.get('sgpr', '')} | {k.get('vgpr_spill', 0)} | "
                     f"{k.get('sgpr_spill', 0)} | {k['lane_hot']} / {k['lane_all']} | {k.get('scratch', 0)} | {k.get('occ', '')} |")
    text = "\n".join(lines) + "\n"
    if args.md:
        Path(args.md).write_text(text)
    spilling = [k for k in ks if k.get("vgpr_spill", 0)]
    print(f"{len(ks)} kernels; {len(spilling)} with VGPR spills; hot-path offenders: {len(bad)}")
    for k in spilling:
        print(f"  spill {k['vgpr_spill']:3d}  vgpr {k.get('vgpr')}  {k['name']}")
    hot_moves = [k for k in hot if k["lane_hot"]]
    print(f"hot-path kernels with SGPR-spill lane moves inside an MFMA block: {len(hot_moves)}")
    for k in hot_moves:
        print(f"  {k['lane_hot']:3d} of {k['lane_all']:3d} lane moves  {k['name']}")
    if args.strict and bad:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
