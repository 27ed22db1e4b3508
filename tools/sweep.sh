#!/bin/bash
# chunk/stream sweep of the sampling chain (diagnostic, not part of the product)
for cfg in "65536 2" "33408 3" "25088 4" "50048 3" "100096 2" "16768 6" "65536 1"; do
  set -- $cfg
  out=$(timeout -k 10 120 python bench.py --patients 100000 --steps 1 --warmup 0 --chunk-rows $1 --streams $2 --no-cpu-baseline --profile-rows 32768 2>/dev/null | tail -1)
  echo "chunk=$1 streams=$2 -> $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["achieved_tflops_end_to_end"])')"
done
