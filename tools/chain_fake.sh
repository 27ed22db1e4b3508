#!/bin/bash
# Timing experiment: chain kernel with its staging DMAs redirected to a small hot region (CHAIN_FAKE_DMA builds in lib_fake{1,2,3}).
set -o pipefail
out=gpurun_out/${1:-fake}; mkdir -p $out
for v in ${VARIANTS:-real fake1 fake2 fake3 real}; do
  lib=$PWD/osteosarcoma_diffusionmodel_amd/lib/libosdiff.so
  [ $v != real ] && lib=$PWD/osteosarcoma_diffusionmodel_amd/lib_$v/libosdiff.so
  OSD_BENCH_TIMING_ONLY=1 OSDIFF_LIB=$lib timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train > $out/bench_$v.json 2> $out/bench_$v.err || { echo "bench $v failed"; tail -5 $out/bench_$v.err; exit 1; }
  python - <<PY
import json; d=json.loads(open('$out/bench_$v.json').read().strip().splitlines()[-1]); print('$v value', d['value'], 'frac', d['roofline']['frac'], 'sampler', d['config']['sampler'])
PY
done
