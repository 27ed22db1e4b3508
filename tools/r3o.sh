#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" OSD_BENCH_ONE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --patients 4096 --steps 1 --warmup 0 --train-steps 20 --no-graph 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['train']['ms_per_step'], d['train']['exposed_comm_ms_per_step'])"; }
run A=1
run OSD_KEEP_TORCH_THREADS=1
run OMP_NUM_THREADS=4
run OSD_COMM_PRIO_DEFAULT=1
