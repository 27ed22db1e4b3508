#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "OSD_WGRAD_ITEMS=512" "OSD_WGRAD_ITEMS=384" "OSD_WGRAD_ITEMS=448" "OSD_WGRAD_ITEMS=640" "OSD_WGRAD_ITEMS=768" "OSD_WGRAD_ITEMS=1024" "OSD_WGRAD_ITEMS=512"; do
  echo "== $v"; env $v timeout -k 10 120 python bench.py --train-only --train-steps 80 2>&1 | tail -1 | cut -c1-130 || exit 1
done
