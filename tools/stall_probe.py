#!/usr/bin/env python3
"""Periodic ~60 ms stalls in a training loop that synchronises every few steps: where do they come from?
Runs the config-2 train_step loop with a synchronisation every k steps (k = 7, 14, 28, none) and lists the positions of slow
steps; then a loop of trivial torch kernels with the same launch count for comparison."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import CONF
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
from osteosarcoma_diffusionmodel_amd.train import Trainer

B = 4096
conf = {"model": dict(CONF["model"])}
conf["training"] = {"learning_rate": 1e-4, "weight_decay": 1e-5, "patience": 100, "min_delta": 1e-4, "augmentation": {"mixup_alpha": 0.2},
                    "save_dir": "/tmp/osd_probe", "num_epochs": 1, "save_frequency": 10, "val_split": 0.2, "random_seed": 42, "batch_size": B}
torch.manual_seed(0)
model = BiologyAwareDiffusionModel(50, 1900, 50, 3, conf)
tr = Trainer(model, [], [], conf, device="cuda")
model.train()
x = torch.randn(B, 2000, device="cuda"); c = torch.randn(B, 3, device="cuda")
for i in range(30):
    tr.train_step(x, c)
torch.cuda.synchronize()


def run(k, steps=140, label=""):
    slow = []
    t0 = time.perf_counter()
    last = t0
    for i in range(steps):
        tr.train_step(x, c)
        if k and (i + 1) % k == 0:
            torch.cuda.synchronize()
            now = time.perf_counter()
            if now - last > (k * 1.0 + 8) * 1e-3:
                slow.append((i + 1, round(1e3 * (now - last), 1)))
            last = now
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label} sync every {k or 'never'}: {1e3 * dt / steps:.3f} ms/step; slow windows (step, ms): {slow}")


for k in (0, 7, 14, 28, 0):
    run(k, label="train_step")
# the same with a sleep after each sync (the GPU idles longer)
def run_idle(k, idle_ms, steps=140):
    slow = []
    tot = 0.0
    for i in range(steps // k):
        t0 = time.perf_counter()
        for _ in range(k):
            tr.train_step(x, c)
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        tot += d
        if d > (k + 8) * 1e-3:
            slow.append((i, round(1e3 * d, 1)))
        time.sleep(idle_ms * 1e-3)
    print(f"train_step sync every {k} + {idle_ms} ms idle: {1e3 * tot / (steps // k * k):.3f} ms/step busy; slow windows: {slow}")


run_idle(14, 0.0)
run_idle(14, 5.0)
run_idle(14, 50.0)
# trivial kernels, same number of launches per window (about 50 per step)
a = torch.zeros(1 << 20, device="cuda")
def run_trivial(k, steps=140):
    slow = []
    for i in range(steps // k):
        t0 = time.perf_counter()
        for _ in range(k * 50):
            a.add_(1.0)
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        if d > 20e-3:
            slow.append((i, round(1e3 * d, 1)))
    print(f"trivial kernels, {k * 50} launches per window: slow windows: {slow}")
run_trivial(14)
