#!/usr/bin/env python3
"""Where a device-resident training epoch spends its time (host vs device): Trainer.train_epoch at the config-2 shape."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import CONF
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
from osteosarcoma_diffusionmodel_amd.train import Trainer, OsteosarcomaDataset

B, rows = 4096, 65536
conf = {"model": dict(CONF["model"])}
conf["training"] = {"learning_rate": 1e-4, "weight_decay": 1e-5, "patience": 100, "min_delta": 1e-4, "augmentation": {"mixup_alpha": 0.2},
                    "save_dir": "/tmp/osd_probe", "num_epochs": 1, "save_frequency": 10, "val_split": 0.2, "random_seed": 42, "batch_size": B}
torch.manual_seed(0)
model = BiologyAwareDiffusionModel(50, 1900, 50, 3, conf)
ds = object.__new__(OsteosarcomaDataset)
g = torch.Generator().manual_seed(1)
ds.data, ds.conditions, ds.survival_days = torch.randn(rows, 2000, generator=g), torch.randn(rows, 3, generator=g), torch.rand(rows, generator=g)
tr_ds, va_ds = torch.utils.data.random_split(ds, [rows - 8192, 8192], generator=torch.Generator().manual_seed(42))
loader = torch.utils.data.DataLoader(tr_ds, batch_size=B, shuffle=True, num_workers=0, drop_last=True)
tr = Trainer(model, loader, loader, conf, device="cuda")
for _ in range(3):
    tr.train_epoch()
torch.cuda.synchronize()
res, _ = tr._resident_splits()
t0 = time.perf_counter(); idx = res.epoch_indices(); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"epoch_indices: {1e3 * (t1 - t0):.2f} ms for {len(idx)} batches")
data, cond, surv = res.base
model.train()
host = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for rep in range(5):
    for ix in idx:
        h0 = time.perf_counter()
        lam, perm = tr.mixup.draw(ix.shape[0], ix.device)
        h1 = time.perf_counter()
        ib = ix[perm]
        h2 = time.perf_counter()
        tr.train_step(None, None, source=(data, cond, surv, ix, ib, lam))
        h3 = time.perf_counter()
        host.append((h1 - h0, h2 - h1, h3 - h2))
t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
n = len(host)
print(f"{n} steps: host loop {1e3 * t_host / n:.3f} ms/step, with device {1e3 * t_all / n:.3f} ms/step")
import numpy as np
h = np.array(host) * 1e3
print("host per step: mixup.draw %.3f ms, idx[perm] %.3f ms, train_step %.3f ms" % tuple(h.mean(0)))
print("host per step (median): %.3f %.3f %.3f" % tuple(np.median(h, 0)))
t0 = time.perf_counter()
for _ in range(5):
    tr.train_epoch()
torch.cuda.synchronize()
print(f"train_epoch: {1e3 * (time.perf_counter() - t0) / 5 / len(idx):.3f} ms/step")
# phases of train_epoch
for rep in range(3):
    torch.cuda.synchronize(); a = time.perf_counter()
    idx = res.epoch_indices(); b = time.perf_counter()
    total = torch.zeros(1, device="cuda")
    for ix in idx:
        lam, perm = tr.mixup.draw(ix.shape[0], ix.device)
        total += tr.train_step(None, None, source=(data, cond, surv, ix, ix[perm], lam))
    c = time.perf_counter()
    v = float(total.item()); d = time.perf_counter()
    print(f"epoch {rep}: indices {1e3*(b-a):.2f} ms, host loop {1e3*(c-b):.2f} ms, final item() wait {1e3*(d-c):.2f} ms")
# per-step host times over 12 epochs: find the outliers
import gc
rec = []
for rep in range(12):
    idx = res.epoch_indices()
    total = torch.zeros(1, device="cuda")
    for k, ix in enumerate(idx):
        h0 = time.perf_counter()
        lam, perm = tr.mixup.draw(ix.shape[0], ix.device)
        h1 = time.perf_counter()
        ib = ix[perm]
        h2 = time.perf_counter()
        l = tr.train_step(None, None, source=(data, cond, surv, ix, ib, lam))
        h3 = time.perf_counter()
        total += l
        h4 = time.perf_counter()
        rec.append((rep, k, h1 - h0, h2 - h1, h3 - h2, h4 - h3))
    v = float(total.item())
for r in rec:
    if max(r[2:]) > 3e-3:
        print("outlier epoch %d step %d: draw %.2f ms, gather %.2f ms, train_step %.2f ms, += %.2f ms" % (r[0], r[1], 1e3*r[2], 1e3*r[3], 1e3*r[4], 1e3*r[5]))
print("gc counts", gc.get_count(), "gc stats", gc.get_stats()[2])
# the same epochs with the per-epoch read-back through a pinned buffer + event polling instead of a blocking .item()
pin = torch.zeros(1).pin_memory()
evd = torch.cuda.Event()
ts = []
for rep in range(12):
    a = time.perf_counter()
    idx = res.epoch_indices()
    total = torch.zeros(1, device="cuda")
    for k, ix in enumerate(idx):
        lam, perm = tr.mixup.draw(ix.shape[0], ix.device)
        total += tr.train_step(None, None, source=(data, cond, surv, ix, ix[perm], lam))
    pin.copy_(total, non_blocking=True)
    evd.record()
    while not evd.query():
        time.sleep(50e-6)
    v = float(pin[0])
    ts.append(1e3 * (time.perf_counter() - a))
print("polled read-back, ms per epoch:", " ".join(f"{t:.1f}" for t in ts))
