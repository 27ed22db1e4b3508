#!/bin/bash
# A/B of the chain kernel's tile variants on one box: bitwise tests with each, then the sampling leg of bench.py.
set -o pipefail
out=gpurun_out/${1:-ab}; mkdir -p $out
for w in 8 4; do
  OSD_CHAIN_WAVES=$w timeout -k 10 300 python -m pytest tests/test_gpu_chain.py -x -q > $out/pytest_w$w.log 2>&1; echo "waves=$w pytest rc=$?"; tail -2 $out/pytest_w$w.log
done
for w in 8 4 8 4; do
  OSD_CHAIN_WAVES=$w timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train > $out/bench_w$w.json 2> $out/bench_w$w.err || { echo "bench waves=$w failed"; tail -5 $out/bench_w$w.err; exit 1; }
  python - <<PY
import json; d=json.loads(open('$out/bench_w$w.json').read().strip().splitlines()[-1]); print('waves=$w value', d['value'], 'frac', d['roofline']['frac'], 'sampler', d['config']['sampler'])
PY
done
