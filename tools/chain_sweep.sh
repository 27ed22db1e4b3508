#!/bin/bash
# chain-kernel knobs at the BASELINE config-3 shape (diagnostic)
run() { timeout -k 10 200 python bench.py --patients 100000 --steps 1 --warmup 0 --no-cpu-baseline --no-train --profile-rows 32768 "$@" 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["config"]["sampler"], d["value"], d["achieved_tflops_end_to_end"])'; }
echo "graph:"; run --sampler graph
for st in 30000 0 15000 60000 120000; do echo "chain stagger=$st:"; run --sampler chain --chain-stagger $st; done
echo "chain, 100 steps per launch:"; run --sampler chain --chain-steps-per-launch 100
