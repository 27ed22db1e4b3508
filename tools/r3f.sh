#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3f
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_pipeline.py tests/test_gpu_ddp.py tests/test_gpu_cvae.py -x -q > gpurun_out/r3f/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/r3f/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --train-only --train-steps 40 2>&1 | tail -1 | tee gpurun_out/r3f/train.json
