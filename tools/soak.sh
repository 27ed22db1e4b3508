#!/bin/bash
# Repeat the things that involve inter-workgroup hand-offs: the GPU suite twice, the squad fuzz, 300 training steps with a finite loss.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/soak
for i in 1 2; do timeout -k 10 400 python -m pytest tests -m gpu -q -x > gpurun_out/soak/pytest_$i.log 2>&1; echo "pytest pass $i rc=$? $(tail -1 gpurun_out/soak/pytest_$i.log)"; done
timeout -k 10 600 python tools/squad_fuzz.py 80 > gpurun_out/soak/fuzz.log 2>&1; echo "fuzz rc=$? $(tail -1 gpurun_out/soak/fuzz.log)"
timeout -k 10 300 python - <<'PY' > gpurun_out/soak/train.log 2>&1
import sys, math, torch
sys.path.insert(0, ".")
from bench import CONF
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
torch.manual_seed(0)
m = BiologyAwareDiffusionModel(50, 1900, 50, 3, CONF).cuda().train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
x = torch.randn(4096, 2000, device="cuda"); c = torch.randn(4096, 3, device="cuda")
bad = 0
for i in range(300):
    opt.zero_grad()
    loss = m(x, c, seed=i)
    loss.backward()
    opt.step()
    if i % 25 == 0 or not math.isfinite(loss.item()):
        print(i, loss.item(), flush=True)
    bad += not math.isfinite(loss.item())
print("non-finite losses:", bad)
PY
echo "train rc=$? $(tail -1 gpurun_out/soak/train.log)"
