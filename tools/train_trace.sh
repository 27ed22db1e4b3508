#!/bin/bash
# Kernel timeline of one training step (config 2) under a precision: train_trace.sh [fp32|bf16x3]  -> gpurun_out/prof_train_<precision>/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
prec=${1:-bf16x3}
out=gpurun_out/prof_train_$prec; rm -rf $out; mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 bench.py --train-only --train-steps 40 --train-precision $prec > $out/log.txt 2>&1; echo rc=$?
f=$(find $out/t -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats.csv
python3 tools/train_timeline.py $out/t > $out/timeline.txt 2>&1
find $out/t -type f -delete
head -45 $out/timeline.txt | cut -c1-150
