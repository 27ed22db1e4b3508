#!/bin/bash
# One GPU call: full parity suite, default bench (one JSON line -> gpurun_out/rc/bench.json), PMC table of the training kernels.
set -o pipefail
mkdir -p gpurun_out/rc
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/rc/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/rc/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
( time timeout -k 10 600 python bench.py > gpurun_out/rc/bench.json 2> gpurun_out/rc/bench.err ) 2>&1 | grep real; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/rc/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'roof', d['roofline']['frac'], 'train', d['train']['ms_per_step'], d['train']['epoch_samples_per_s'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print('refw', [(r['D'], r['patients_per_scenario'], r['patients_per_s']) for r in d['reference_workload']['runs']])
print('validate', [(s['mmd']['tflops'], s['validate_all']['s']) for s in d['validate']['scenarios']])
PY
bash tools/train_pmc.sh r04 > gpurun_out/rc/train_pmc.log 2>&1; tail -3 gpurun_out/rc/train_pmc.log
