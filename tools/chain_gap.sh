#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/gap; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/tr -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train > $out/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/gap/tr/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ch = [i for i, r in enumerate(rows) if "chain_kernel" in r["Kernel_Name"]]
print("chain launches:", len(ch))
for a, b in zip(ch[:-1], ch[1:]):
    e0 = int(rows[a]["End_Timestamp"]); s1 = int(rows[b]["Start_Timestamp"])
    print(f"gap between chain kernels: {(s1 - e0) / 1e6:.2f} ms; kernels in between:")
    for r in rows[a + 1:b]:
        print(f"   +{(int(r['Start_Timestamp']) - e0) / 1e6:8.2f} ms  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  {r['Kernel_Name'][:90]}")
m = sorted(glob.glob("gpurun_out/gap/tr/**/*memory_copy_trace.csv", recursive=True))
if m:
    mr = sorted(csv.DictReader(open(m[-1])), key=lambda r: int(r["Start_Timestamp"]))
    e0 = int(rows[ch[-2]]["End_Timestamp"]); s1 = int(rows[ch[-1]]["Start_Timestamp"])
    for r in mr:
        s = int(r["Start_Timestamp"])
        if e0 <= s <= s1: print(f"   copy +{(s - e0) / 1e6:8.2f} ms {(int(r['End_Timestamp']) - s) / 1e3:9.1f} us {r.get('Direction', '')} {r.get('Bytes', r.get('Size', ''))}")
PY
rm -rf $out/tr
tail -1 $out/log.txt | cut -c1-200
