#!/usr/bin/env python3
"""Feasibility probe: is one training step of B rows slower than two concurrent steps of B/2 rows on two streams (two
models / handles)?  If the pair is clearly faster, running the halves of a batch on two streams inside one step pays."""
import sys, time, threading
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from bench import CONF
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
from osteosarcoma_diffusionmodel_amd.train import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
conf = dict(CONF)
conf["training"] = {"learning_rate": 1e-4, "weight_decay": 1e-5, "patience": 100, "min_delta": 1e-4, "augmentation": {"mixup_alpha": 0.2},
                    "save_dir": "/tmp/osd_ckpt", "num_epochs": 1, "save_frequency": 10, "val_split": 0.2, "random_seed": 42, "batch_size": B}
g = torch.Generator(device="cuda").manual_seed(42)
data = torch.randn(65536, 2000, device="cuda", generator=g); cond = torch.randn(65536, 3, device="cuda", generator=g)
def make():
    torch.manual_seed(0)
    m = BiologyAwareDiffusionModel(50, 1900, 50, 3, conf); t = Trainer(m, [], [], conf, device="cuda"); m.train(); return t
def run(trs, rows, steps):
    streams = [torch.cuda.Stream() for _ in trs]
    def one(i):
        for t, s in zip(trs, streams):
            with torch.cuda.stream(s):
                o = (i * rows) % (65536 - rows)
                t.train_step(data[o:o + rows], cond[o:o + rows])
    for i in range(20): one(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): one(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps
a = run([make()], B, 60)
b = run([make(), make()], B // 2, 60)
c = run([make()], B // 2, 60)
print(f"one step of {B}: {a*1e3:.3f} ms; two concurrent steps of {B//2} on two streams: {b*1e3:.3f} ms; one step of {B//2}: {c*1e3:.3f} ms")
