#!/bin/bash
# One GPU call: full parity suite, default bench, rocprofv3 summaries, training bench.  Writes under gpurun_out/rc/.
set -o pipefail
mkdir -p gpurun_out/rc
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/rc/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/rc/summary.txt
tail -3 gpurun_out/rc/pytest_gpu.log | tee -a gpurun_out/rc/summary.txt
for t in 256 512 1024; do OSD_WGRAD_TARGET=$t timeout -k 10 120 python tools/train_bench.py 4096 50 2>&1 | tail -1 | sed "s/^/wgrad_target=$t /" | tee -a gpurun_out/rc/summary.txt; done
timeout -k 10 120 python tools/train_bench.py 16384 20 2>&1 | tail -1 | tee -a gpurun_out/rc/summary.txt
timeout -k 10 400 python bench.py > gpurun_out/rc/bench.json 2> gpurun_out/rc/bench.err; echo "bench rc=$?" | tee -a gpurun_out/rc/summary.txt
python -c "
import json; d=json.loads(open('gpurun_out/rc/bench.json').read().strip().splitlines()[-1]); print('value', d['value'], 'e2e TF', d['achieved_tflops_end_to_end'], 'roof', d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])" | tee -a gpurun_out/rc/summary.txt
