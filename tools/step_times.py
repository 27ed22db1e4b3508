#!/usr/bin/env python3
"""Per-step device time of the training loop (event pair per step) over a long run: where are the slow steps?"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from bench import CONF
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
from osteosarcoma_diffusionmodel_amd.train import Trainer

B, rows = 4096, 65536
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
mode = sys.argv[2] if len(sys.argv) > 2 else "full"      # full | nodraw (fixed indices, no host draws) | nogc
conf = {"model": dict(CONF["model"])}
conf["training"] = {"learning_rate": 1e-4, "weight_decay": 1e-5, "patience": 100, "min_delta": 1e-4, "augmentation": {"mixup_alpha": 0.2},
                    "save_dir": "/tmp/osd_probe", "num_epochs": 1, "save_frequency": 10, "val_split": 0.2, "random_seed": 42, "batch_size": B}
torch.manual_seed(0)
model = BiologyAwareDiffusionModel(50, 1900, 50, 3, conf)
tr = Trainer(model, [], [], conf, device="cuda")
model.train()
data = torch.randn(rows, 2000, device="cuda"); cond = torch.randn(rows, 3, device="cuda"); surv = torch.rand(rows, device="cuda")
order = torch.arange(rows, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
host = []
import gc
if mode == "nogc":
    gc.disable()
fixed_perm = torch.randperm(B, device="cuda")
lams, perms, _ = tr.mixup.draw_epoch([B] * steps, "cuda")
reserved = []
torch.cuda.synchronize()
t0 = time.perf_counter()
ev[0].record()
for i in range(steps):
    h0 = time.perf_counter()
    o = (i * B) % (rows - B)
    idx = order[o:o + B]
    if mode == "nodraw":
        tr.train_step(None, None, source=(data, cond, surv, idx, fixed_perm, 0.3))
    elif mode == "epochdraw":
        tr.train_step(None, None, source=(data, cond, surv, idx, idx[perms[i]], lams[i]))
    else:
        lam, perm = tr.mixup.draw(B, idx.device)
        tr.train_step(None, None, source=(data, cond, surv, idx, idx[perm], lam))
    reserved.append(torch.cuda.memory_reserved())
    ev[i + 1].record()
    host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
d = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])
h = 1e3 * np.array(host)
print(f"{steps} steps: wall {1e3 * wall / steps:.3f} ms/step; device per step: median {np.median(d):.3f}, mean {d.mean():.3f}, p90 {np.percentile(d, 90):.3f}, max {d.max():.2f} ms")
print(f"host per step: median {np.median(h):.3f}, mean {h.mean():.3f}, max {h.max():.2f} ms")
for lo in range(0, steps, 20):
    print(f"steps {lo:3d}-{lo + 19:3d}: device mean {d[lo:lo + 20].mean():.3f} ms, host mean {h[lo:lo + 20].mean():.3f} ms")
print("mode", mode, "memory_reserved changes at steps:", [i for i in range(1, steps) if reserved[i] != reserved[i - 1]][:20], "gc", gc.get_count())
slow = [(i, round(float(d[i]), 2)) for i in range(steps) if d[i] > 1.5 * np.median(d)]
print("slow steps (device):", slow[:30])
