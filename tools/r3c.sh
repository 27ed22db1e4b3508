#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "OSD_BWD_FLAGS=0" "OSD_BWD_FLAGS=1" "OSD_BWD_FLAGS=2" "OSD_BWD_FLAGS=3" "OSD_BWD_FLAGS=3 OSD_BWD_GRID=256"; do
  echo "== $v"; env $v timeout -k 10 120 python bench.py --train-only --train-steps 60 2>&1 | tail -1 | cut -c1-140 || exit 1
done
