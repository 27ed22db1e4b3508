#!/bin/bash
# One GPU call at round end: rocprofv3 kernel-trace summaries of the final build (copied to profiles/ afterwards).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_final
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench100k -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/bench100k.log 2>&1; echo "bench100k rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/profile_only -- python3 bench.py --profile-only > $out/profile_only.log 2>&1; echo "profile_only rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/train -- python3 tools/train_bench.py 4096 20 > $out/train.log 2>&1; echo "train rc=$?"
for d in bench100k profile_only train; do f=$(find $out/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/${d}_kernel_stats.csv; done
# keep only the summaries (traces are large)
find $out -mindepth 2 -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
ls -la $out; tail -1 $out/bench100k.log | cut -c1-300; tail -1 $out/profile_only.log | cut -c1-200; tail -1 $out/train.log
