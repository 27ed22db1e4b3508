#!/bin/bash
# One GPU call: training parity tests, then the fp32 step's timeline and the train-only bench.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/cb
timeout -k 10 420 python -m pytest tests/test_gpu_config2.py tests/test_gpu_train.py -m gpu -x -q > gpurun_out/cb/pytest.log 2>&1; rc=$?; tail -5 gpurun_out/cb/pytest.log
[ $rc -ne 0 ] && exit 1
bash tools/train_trace.sh fp32 && timeout -k 10 200 python bench.py --train-only --train-steps 200 > gpurun_out/cb/bench.json 2> gpurun_out/cb/bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/cb/bench.json
