#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3d
i=0
for v in "OSD_BWD_FLAGS=0" "OSD_BWD_FLAGS=3" "OSD_BWD_FLAGS=4" "OSD_BWD_FLAGS=7" "OSD_BWD_FLAGS=8" "OSD_BWD_FLAGS=11" "OSD_BWD_FLAGS=11 OSD_BWD_WG_ITEMS=1500" "OSD_BWD_MODE=1 OSD_BWD_FLAGS=3"; do
  i=$((i+1))
  echo "== $v"; env $v OSD_BWD_STAMPS=gpurun_out/r3d/st$i.txt timeout -k 10 120 python bench.py --train-only --train-steps 30 2>&1 | tail -1 | cut -c1-140 || exit 1
  [ -f gpurun_out/r3d/st$i.txt ] && python3 tools/bwd_stamps.py gpurun_out/r3d/st$i.txt 2.4e9 | head -6
done
