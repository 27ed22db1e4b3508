#!/usr/bin/env python3
"""Forward-only training passes (loss, no gradients) at a given batch size: a target for `rocprofv3 --kernel-trace --stats`
(how does train_squad_fwd_kernel's duration scale with the batch?).   train_fwd_sizes.py <rows> [iters]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from bench import CONF
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
n = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.manual_seed(0)
m = BiologyAwareDiffusionModel(50, 1900, 50, 3, CONF).cuda().train()
x = torch.randn(n, 2000, device="cuda"); c = torch.randn(n, 3, device="cuda")
with torch.no_grad():
    for i in range(iters):
        loss = m(x, c, seed=i)
torch.cuda.synchronize()
print("ok", n, float(loss))
