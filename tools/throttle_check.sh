#!/bin/bash
# is the process being CFS-throttled?  cpu.stat before / after tools/epoch_probe.py, default threads vs OMP_NUM_THREADS=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
stat() { cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -E "nr_periods|nr_throttled|throttled_usec" | tr '\n' ' '; echo; }
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; python3 -c "import torch; print('torch threads', torch.get_num_threads())"
echo "before: $(stat)"
timeout -k 10 200 python tools/epoch_probe.py 2>&1 | grep -E "train_epoch:|polled|^epoch [0-9]"
echo "after default: $(stat)"
OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 timeout -k 10 200 python tools/epoch_probe.py 2>&1 | grep -E "train_epoch:|polled|^epoch [0-9]"
echo "after OMP=1: $(stat)"
