#!/usr/bin/env python3
"""Diagnostic: wall time of consecutive groups of training steps (config 2) -- is the bench's training leg at steady state?"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
from osteosarcoma_diffusionmodel_amd.train import Trainer
B = bench.TRAIN_BATCH
conf = {"model": dict(bench.CONF["model"])}
conf["training"] = {"learning_rate": 1e-4, "weight_decay": 1e-5, "patience": 100, "min_delta": 1e-4, "augmentation": {"mixup_alpha": 0.2},
                    "save_dir": "/tmp/osd_bench_ckpt", "num_epochs": 1, "save_frequency": 10, "val_split": 0.2, "random_seed": 42, "batch_size": B}
torch.manual_seed(0)
dev = torch.device("cuda:0")
model = BiologyAwareDiffusionModel(50, 1900, 50, 3, conf)
tr = Trainer(model, [], [], conf, device=dev)
model.train()
g = torch.Generator(device=dev).manual_seed(42)
rows = 65536
data = torch.randn(rows, 2000, device=dev, generator=g); cond = torch.randn(rows, 3, device=dev, generator=g); surv = torch.rand(rows, device=dev, generator=g)
def one(i):
    o = (i * B) % (rows - B); sl = slice(o, o + B)
    mixed = tr.mixup({"data": data[sl], "conditions": cond[sl], "survival": surv[sl]})
    return tr.train_step(mixed["data"], mixed["conditions"])
group = int(sys.argv[1]) if len(sys.argv) > 1 else 5
torch.cuda.synchronize()
for gi in range(16):
    t0 = time.perf_counter()
    for i in range(group): one(gi * group + i)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"steps {gi*group:3d}..{gi*group+group-1:3d}: {dt/group*1e3:.3f} ms/step (host enqueue {th/group*1e3:.3f} ms/step)", flush=True)
