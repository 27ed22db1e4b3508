#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OSD_TRAIN_KSPLIT=1 timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_config2.py -x -q 2>&1 | tail -3
for v in "OSD_TRAIN_KSPLIT=0" "OSD_TRAIN_KSPLIT=1" "OSD_TRAIN_KSPLIT=0" "OSD_TRAIN_KSPLIT=1"; do
  echo "== $v"; env $v timeout -k 10 120 python bench.py --train-only --train-steps 80 2>&1 | tail -1 | cut -c1-130 || exit 1
done
OSD_TRAIN_KSPLIT=1 bash tools/train_tl.sh > /dev/null 2>&1; sed -n 1,19p gpurun_out/tl1/timeline.txt
