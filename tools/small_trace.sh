#!/bin/bash
# Kernel-trace summary of one small-batch reverse chain per engine: small_trace.sh <rows> <T> <engine...>  -> gpurun_out/prof_small/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rows=${1:-3000}; T=${2:-100}; shift 2
out=gpurun_out/prof_small
rm -rf $out; mkdir -p $out
for e in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$e -- python3 tools/small_batch.py $rows $T $e > $out/$e.log 2>&1; echo "$e rc=$?"
  f=$(find $out/$e -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/${e}_${rows}_kernel_stats.csv
  grep rows $out/$e.log
done
find $out -mindepth 2 -type f -delete
python3 - "$out" <<'PY'
import csv, glob, re, sys
for f in sorted(glob.glob(sys.argv[1] + "/*_kernel_stats.csv")):
    print("==", f)
    for r in list(csv.DictReader(open(f)))[:10]:
        n = re.sub(r"\(.*$", "", r["Name"].replace("osd::", ""))
        print("%-100s calls %6s avg %8.1f us total %8.1f ms" % (n[:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
