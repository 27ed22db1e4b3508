#!/bin/bash
# Training-step check on one box: parity tests (single process, 2-rank DDP on one device), then the training leg of bench.py.
set -o pipefail
out=gpurun_out/${1:-tab}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_config2.py tests/test_gpu_ddp.py tests/test_gpu_hygiene.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for i in 1 2 3; do timeout -k 10 120 python bench.py --train-only --train-steps 60 2>&1 | tail -1 | cut -c1-160; done
