#!/usr/bin/env python3
"""Secondary metric (BASELINE config 2): diffusion training samples/s at B=4096, D=2000 on one GPU
(mixup + fused fwd/bwd + clip + AdamW per step)."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from bench import CONF
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
from osteosarcoma_diffusionmodel_amd.train import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
conf = dict(CONF)
conf["training"] = {"learning_rate": 1e-4, "weight_decay": 1e-5, "patience": 100, "min_delta": 1e-4,
                    "augmentation": {"mixup_alpha": 0.2}, "save_dir": "/tmp/osd_ckpt", "num_epochs": 1,
                    "save_frequency": 10, "val_split": 0.2, "random_seed": 42, "batch_size": B}
torch.manual_seed(0)
model = BiologyAwareDiffusionModel(50, 1900, 50, 3, conf)
import os
if os.environ.get("OSD_TRAIN_STREAMS"):
    model.train_streams = int(os.environ["OSD_TRAIN_STREAMS"])
tr = Trainer(model, [], [], conf, device="cuda")
model.train()
g = torch.Generator(device="cuda").manual_seed(42)
data = torch.randn(65536, 2000, device="cuda", generator=g)
data[:, :50] = (data[:, :50] > 0).float()
cond = torch.randn(65536, 3, device="cuda", generator=g)
surv = torch.rand(65536, device="cuda", generator=g)


def one(i):
    sl = slice((i * B) % (65536 - B), (i * B) % (65536 - B) + B)
    mixed = tr.mixup({"data": data[sl], "conditions": cond[sl], "survival": surv[sl]})
    return tr.train_step(mixed["data"], mixed["conditions"])


for i in range(3):
    one(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    loss = one(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"B={B}: {dt*1e3:.3f} ms/step, {B/dt:.0f} samples/s, {14.56e6*B/dt/1e12:.1f} TFLOP/s algorithmic, loss {loss.item():.4f}")
