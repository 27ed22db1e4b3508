#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_config2.py tests/test_gpu_constraints.py -x -q 2>&1 | tail -3
for i in 1 2; do timeout -k 10 120 python bench.py --train-only --train-steps 80 2>&1 | tail -1 | cut -c1-130; done
bash tools/train_tl.sh > /dev/null 2>&1; sed -n 1,7p gpurun_out/tl1/timeline.txt
