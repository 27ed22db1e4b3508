#!/bin/bash
# rocprofv3 kernel-trace summaries of the final build (copied to profiles/ afterwards).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_final
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench100k -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train --no-validate --no-reference-workload --no-mid-size --no-split > $out/bench100k.log 2>&1; echo "bench100k (chain engine) rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench100k_graph -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train --no-validate --no-reference-workload --no-mid-size --no-split --sampler graph > $out/bench100k_graph.log 2>&1; echo "bench100k_graph rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_mid -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train --no-validate --no-reference-workload --no-split > $out/bench_mid.log 2>&1; echo "bench_mid (with the 32 768-patient leg: LDS-resident chain) rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/train -- python3 bench.py --train-only --train-steps 40 --no-split > $out/train.log 2>&1; echo "train rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/train_bf16x3 -- python3 bench.py --train-only --train-steps 40 --train-precision bf16x3 > $out/train_bf16x3.log 2>&1; echo "train (weight gradients on the bf16 pipe) rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/validate -- python3 bench.py --validate-only > $out/validate.log 2>&1; echo "validate rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/refw -- python3 bench.py --reference-workload-only > $out/refw.log 2>&1; echo "refw rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/split -- python3 tools/split_bench.py 100000 100 65536 2 0 > $out/split.log 2>&1; echo "split (bf16x3 engine, 100 000 patients x 100 steps) rc=$?"
for d in bench100k bench100k_graph bench_mid train train_bf16x3 validate refw split; do f=$(find $out/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/${d}_kernel_stats.csv; done
python3 tools/train_timeline.py $out/train > $out/train_timeline.txt 2>&1
python3 tools/train_timeline.py $out/train_bf16x3 > $out/train_bf16x3_timeline.txt 2>&1
find $out -mindepth 2 -type f -delete 2>/dev/null
ls -la $out; tail -1 $out/bench100k.log | cut -c1-300; tail -1 $out/train.log | cut -c1-200
