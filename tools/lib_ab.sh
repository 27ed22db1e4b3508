#!/bin/bash
# Same-box A/B of two builds of the library on the sampling leg of bench.py: lib_ab.sh <outdir> <libdir> [<libdir> ...] (names under osteosarcoma_diffusionmodel_amd/)
set -o pipefail
out=gpurun_out/$1; shift; mkdir -p $out
for rep in 1 2; do for v in "$@"; do
  OSD_BENCH_TIMING_ONLY=1 OSDIFF_LIB=$PWD/osteosarcoma_diffusionmodel_amd/$v/libosdiff.so timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train > $out/bench_$v.json 2> $out/bench_$v.err || { echo "bench $v failed"; tail -5 $out/bench_$v.err; exit 1; }
  python - <<PY
import json; d=json.loads(open('$out/bench_$v.json').read().strip().splitlines()[-1]); print('$v value', d['value'], 'frac', d['roofline']['frac'], 'sampler', d['config']['sampler'])
PY
done; done
