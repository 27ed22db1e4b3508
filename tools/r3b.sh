#!/bin/bash
set -o pipefail
out=gpurun_out/r3b; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -m pytest tests/test_gpu_train.py tests/test_gpu_config2.py -x -q > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $out/pytest.log
[ $rc -ne 0 ] && exit 1
for v in "OSD_PERSISTENT_BWD=0" "OSD_PERSISTENT_BWD=1" "OSD_PERSISTENT_BWD=1 OSD_BWD_WG_ITEMS=512" "OSD_PERSISTENT_BWD=1 OSD_BWD_WG_ITEMS=1200"; do
  echo "== $v"; env $v timeout -k 10 120 python bench.py --train-only --train-steps 60 2>&1 | tail -1 | cut -c1-140 || exit 1
done
rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 bench.py --train-only --train-steps 12 > $out/log.txt 2>&1
python3 tools/train_timeline.py $out/tr > $out/timeline.txt 2>&1
find $out/tr -type f -delete
cat $out/timeline.txt
