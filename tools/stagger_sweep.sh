#!/bin/bash
for st in 30000 0 10000 60000 120000 30000; do
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --chain-stagger $st > /tmp/b.json 2>/tmp/b.err || { echo fail; tail -3 /tmp/b.err; exit 1; }
  python - <<PY
import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('stagger $st value', d['value'], 'frac', d['roofline']['frac'])
PY
done
