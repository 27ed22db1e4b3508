#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3e
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_chain.py tests/test_gpu_config2.py -x -q -k "not config3_full" > gpurun_out/r3e/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/r3e/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --reference-workload-only 2>&1 | tail -1 | tee gpurun_out/r3e/refw.json; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
