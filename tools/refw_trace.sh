#!/bin/bash
# kernel trace of the reference's default generation workload (bench.py --reference-workload-only): per-kernel averages + idle share
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/refw; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -- python3 bench.py --reference-workload-only > $out/log.txt 2>&1; echo "rc=$?"
f=$(find $out/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/kernel_stats.csv
t=$(find $out/tr -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY' | tee $out/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 40 000 kernels ~ the D = 2000 runs; take a window from the first third (D = 5142, 1000 patients per scenario)
n = len(rows)
w = rows[n // 10: n // 10 + 14 * 200]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in w)
span = int(w[-1]["End_Timestamp"]) - int(w[0]["Start_Timestamp"])
print(f"window: {len(w)} kernels, span {span/1e3:.1f} us, busy {busy/1e3:.1f} us ({busy/span:.2%}), mean gap {(span-busy)/len(w)/1e3:.2f} us")
agg = collections.defaultdict(list)
for r in w:
    agg[r["Kernel_Name"][:110]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v)/len(v):8.2f} us x {len(v):5d}  {k}")
PY
tail -1 $out/log.txt | cut -c1-400
find $out/tr -type f -delete
