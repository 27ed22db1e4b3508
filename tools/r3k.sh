#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3k
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_config2.py -x -q > gpurun_out/r3k/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r3k/pytest.log
[ $rc -ne 0 ] && exit 1
for v in "OSD_WGRAD_XCD=0" "OSD_WGRAD_XCD=1" "OSD_WGRAD_XCD=0" "OSD_WGRAD_XCD=1"; do
  echo "== $v"; env $v timeout -k 10 120 python bench.py --train-only --train-steps 60 2>&1 | tail -1 | cut -c1-150 || exit 1
done
OSD_WGRAD_XCD=1 bash tools/train_tl.sh > /dev/null 2>&1; grep "wgrad_group\|step:" gpurun_out/tl1/timeline.txt
OSD_WGRAD_XCD=0 bash tools/train_tl.sh > /dev/null 2>&1; grep "wgrad_group\|step:" gpurun_out/tl1/timeline.txt
