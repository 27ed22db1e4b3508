#!/usr/bin/env python3
"""Bisect the 65 ms stalls of the epoch path: variants of a 14-step window loop."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from bench import CONF
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
from osteosarcoma_diffusionmodel_amd.train import Trainer

B, rows = 4096, 65536
conf = {"model": dict(CONF["model"])}
conf["training"] = {"learning_rate": 1e-4, "weight_decay": 1e-5, "patience": 100, "min_delta": 1e-4, "augmentation": {"mixup_alpha": 0.2},
                    "save_dir": "/tmp/osd_probe", "num_epochs": 1, "save_frequency": 10, "val_split": 0.2, "random_seed": 42, "batch_size": B}
torch.manual_seed(0)
model = BiologyAwareDiffusionModel(50, 1900, 50, 3, conf)
tr = Trainer(model, [], [], conf, device="cuda")
model.train()
data = torch.randn(rows, 2000, device="cuda"); cond = torch.randn(rows, 3, device="cuda"); surv = torch.rand(rows, device="cuda")
x = data[:B].clone(); c = cond[:B].clone()
order = torch.randperm(rows, device="cuda")
for i in range(30):
    tr.train_step(x, c)
torch.cuda.synchronize()


def windows(name, body, n_win=24, k=14):
    slow, tot = [], 0.0
    for w in range(n_win):
        t0 = time.perf_counter()
        total = torch.zeros(1, device="cuda")
        for i in range(k):
            total += body(w, i)
        v = float(total.item())
        d = time.perf_counter() - t0
        tot += d
        if d > (k + 10) * 1e-3:
            slow.append((w, round(1e3 * d)))
    print(f"{name}: {1e3 * tot / (n_win * k):.3f} ms/step, slow windows {slow}", flush=True)


windows("A plain train_step(x, c)", lambda w, i: tr.train_step(x, c))
windows("B mixup() + train_step (bench.py's step)", lambda w, i: (lambda m: tr.train_step(m["data"], m["conditions"]))(tr.mixup({"data": x, "conditions": c, "survival": surv[:B]})))
def body_c(w, i):
    ix = order[i * B:(i + 1) * B]
    return tr.train_step(None, None, source=(data, cond, surv, ix, None, 1.0))
windows("C batch source, no mixup", body_c)
def body_d(w, i):
    ix = order[i * B:(i + 1) * B]
    lam, perm = tr.mixup.draw(B, ix.device)
    return tr.train_step(None, None, source=(data, cond, surv, ix, ix[perm], lam))
windows("D batch source + mixup.draw", body_d)
def body_e(w, i):
    ix = order[i * B:(i + 1) * B]
    perm = torch.randperm(B, device="cuda")
    return tr.train_step(None, None, source=(data, cond, surv, ix, ix[perm], 0.3))
windows("E batch source + device randperm", body_e)
windows("A again", lambda w, i: tr.train_step(x, c))
