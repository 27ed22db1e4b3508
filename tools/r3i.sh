#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3i
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_config2.py tests/test_gpu_ddp.py tests/test_gpu_constraints.py tests/test_gpu_model.py -x -q > gpurun_out/r3i/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/r3i/pytest.log
[ $rc -ne 0 ] && exit 1
for v in "OSD_DUAL_DGRAD=0" "OSD_DUAL_DGRAD=1" "OSD_DUAL_DGRAD=1 OSD_TRAIN_INPUT_SPLITK=2" "OSD_DUAL_DGRAD=1 OSD_TRAIN_INPUT_SPLITK=4"; do
  echo "== $v"; env $v timeout -k 10 120 python bench.py --train-only --train-steps 60 2>&1 | tail -1 | cut -c1-200 || exit 1
done
echo "== refw after sampling only"; timeout -k 10 300 python bench.py --no-train --no-validate --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print([(r['D'], r['patients_per_scenario'], r['patients_per_s']) for r in d['reference_workload']['runs']])"
echo "== refw after sampling + train"; timeout -k 10 300 python bench.py --no-validate --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print([(r['D'], r['patients_per_scenario'], r['patients_per_s']) for r in d['reference_workload']['runs']])"
