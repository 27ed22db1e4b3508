#!/bin/bash
# A/B on one box: the conditioning branch's backward as one launch vs four, with and without the side stream beside it.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/cb
for cfg in "1 1" "0 1" "1 0" "0 0"; do
  set -- $cfg
  export OSD_COND_BWD_FUSED=$1 OSD_TWO_STREAM_BWD=$2
  out=gpurun_out/cb/f$1_s$2; rm -rf $out; mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 bench.py --train-only --train-steps 40 > $out/log.txt 2>&1 || exit 1
  python3 tools/train_timeline.py $out/t > $out/timeline.txt 2>&1; find $out/t -type f -delete
  echo "== fused=$1 two_stream=$2"; sed -n 1p $out/timeline.txt; sed -n '/train_squad_bwd/,/wgrad_group_kernel/p' $out/timeline.txt | cut -c1-120
  timeout -k 10 200 python bench.py --train-only --train-steps 300 2>/dev/null | python3 -c "import sys,json; print('ms_per_step', json.loads(sys.stdin.read().strip().splitlines()[-1])['train']['ms_per_step'])" || exit 1
done
