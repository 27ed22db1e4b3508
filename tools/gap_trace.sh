#!/bin/bash
# kernel trace of tools/epoch_probe.py: the longest kernels and the largest gaps between consecutive kernels
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/gap; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/tr -- python3 tools/epoch_probe.py > $out/log.txt 2>&1; echo "rc=$?"
t=$(find $out/tr -name "*kernel_trace.csv" | head -1)
m=$(find $out/tr -name "*memory_copy_trace.csv" | head -1)
python3 - "$t" "$m" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
print("kernels:", len(rows))
long = sorted(rows, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)[:6]
for r in long:
    print(f"long kernel {(int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3:10.1f} us at {(int(r['Start_Timestamp'])-t0)/1e6:9.2f} ms  {r['Kernel_Name'][:70]}")
gaps = []
end = int(rows[0]["End_Timestamp"])
for a, b in zip(rows, rows[1:]):
    end = max(end, int(a["End_Timestamp"]))
    g = int(b["Start_Timestamp"]) - end
    gaps.append((g, a, b))
gaps.sort(key=lambda x: -x[0])
for g, a, b in gaps[:12]:
    print(f"gap {g/1e3:10.1f} us at {(int(b['Start_Timestamp'])-t0)/1e6:9.2f} ms  after {a['Kernel_Name'][:40]} before {b['Kernel_Name'][:40]}")
if len(sys.argv) > 2 and sys.argv[2]:
    mc = list(csv.DictReader(open(sys.argv[2])))
    print("memory copies:", len(mc), mc[0].keys() if mc else "")
    mc.sort(key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)
    for r in mc[:8]:
        print(f"long copy {(int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3:10.1f} us at {(int(r['Start_Timestamp'])-t0)/1e6:9.2f} ms {r.get('Direction','')} {r.get('Size','')}")
PY
find $out/tr -type f -delete
