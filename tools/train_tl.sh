#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tl1; rm -rf $out; mkdir -p $out
export OSD_BWD_CU_SPLIT=${1:-1}
rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 bench.py --train-only --train-steps 30 > $out/log.txt 2>&1
python3 tools/train_timeline.py $out/tr > $out/timeline.txt 2>&1
find $out/tr -type f -delete
tail -1 $out/log.txt | cut -c1-150
