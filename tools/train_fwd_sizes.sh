#!/bin/bash
# train_squad_fwd_kernel's duration over batch sizes (kernel-trace stats of forward-only passes) -> gpurun_out/prof_fwd_sizes/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_fwd_sizes; rm -rf $out; mkdir -p $out
for n in 2048 4096 8192; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/n$n -- python3 tools/probes/train_fwd_sizes.py $n 20 > $out/n$n.log 2>&1; echo "n=$n rc=$?"
  f=$(find $out/n$n -name "*kernel_stats.csv" | head -1); grep "train_squad\|EpiInput\|EpiMse\|q_sample" "$f" | awk -F, '{printf "   %-60s avg %8.1f us\n", substr($1,1,60), $4/1000}'
  find $out/n$n -type f -delete
done
