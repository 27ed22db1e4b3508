#!/usr/bin/env python3
"""Why is hipGraph-replayed small-batch sampling 3x slower after the training leg of bench.py?  Time generate(1000) at D = 2000
after each of several candidate culprits."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import CONF, SCENARIOS
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, SyntheticPatientGenerator
from osteosarcoma_diffusionmodel_amd.train import Trainer, OsteosarcomaDataset

torch.manual_seed(0)
gm = BiologyAwareDiffusionModel(50, 1900, 50, 3, CONF).cuda().eval()
gen = SyntheticPatientGenerator(gm, CONF, device="cuda")


def t_gen(label):
    gen.generate(128, SCENARIOS[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gen.generate(1000, SCENARIOS[0])
    torch.cuda.synchronize()
    print(f"{label}: generate(1000) {time.perf_counter() - t0:.3f} s", flush=True)


t_gen("fresh")
B = 4096
conf = {"model": dict(CONF["model"])}
conf["training"] = {"learning_rate": 1e-4, "weight_decay": 1e-5, "patience": 100, "min_delta": 1e-4, "augmentation": {"mixup_alpha": 0.2},
                    "save_dir": "/tmp/osd_probe", "num_epochs": 1, "save_frequency": 10, "val_split": 0.2, "random_seed": 42, "batch_size": B}
m = BiologyAwareDiffusionModel(50, 1900, 50, 3, conf)
tr = Trainer(m, [], [], conf, device="cuda")
t_gen("after Trainer construction")
m.train()
x = torch.randn(B, 2000, device="cuda"); c = torch.randn(B, 3, device="cuda")
m.train_streams = 1
for _ in range(5):
    tr.train_step(x, c)
torch.cuda.synchronize()
t_gen("after 5 train steps, one stream")
m.train_streams = 2
m._engine()
for _ in range(5):
    tr.train_step(x, c)
torch.cuda.synchronize()
t_gen("after 5 train steps, two streams")
rows = 65536
ds = object.__new__(OsteosarcomaDataset)
g = torch.Generator().manual_seed(1)
ds.data, ds.conditions, ds.survival_days = torch.randn(rows, 2000, generator=g), torch.randn(rows, 3, generator=g), torch.rand(rows, generator=g)
t_gen("after creating a 0.5 GB CPU dataset")
tr_ds, va_ds = torch.utils.data.random_split(ds, [rows - 8192, 8192], generator=torch.Generator().manual_seed(42))
tr.train_loader = torch.utils.data.DataLoader(tr_ds, batch_size=B, shuffle=True, num_workers=0, drop_last=True)
tr.val_loader = torch.utils.data.DataLoader(va_ds, batch_size=B, shuffle=False, num_workers=0)
tr.resident = None
tr.train_epoch(); torch.cuda.synchronize()
t_gen("after one resident epoch")
tr.resident = False
it = iter(tr.train_loader)
batch = next(it)
t_gen("after collating ONE DataLoader batch on the host")
d = batch["data"].to("cuda"); torch.cuda.synchronize()
t_gen("after one pageable 32 MB host-to-device copy")
tr.train_epoch(); torch.cuda.synchronize()
t_gen("after one DataLoader epoch")
import os
print("threads:", torch.get_num_threads(), "cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)
time.sleep(2.0)
t_gen("... and 2 s of sleep later")
import gc
del tr, m
gc.collect(); torch.cuda.empty_cache()
t_gen("after deleting the Trainer and its model")
