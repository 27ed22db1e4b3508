#!/usr/bin/env python3
"""Reverse chain under the bf16x3 split precision vs the fp32 engines at a chosen row count / T (patient-steps/s).
   split_bench.py <rows> <T> [chunk_rows,chunk_rows,...] [streams,streams,...] [fp32: 0|1]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import CONF, scenario_conditions
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
chunks = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [65536]
streams = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [2]
do_fp32 = int(sys.argv[5]) if len(sys.argv) > 5 else 1
conf = {"model": dict(CONF["model"])}; conf["model"]["diffusion"] = {"num_steps": steps, "beta_schedule": "cosine"}
torch.manual_seed(0)
m = BiologyAwareDiffusionModel(50, 1900, 50, 3, conf).cuda().eval()
cond = scenario_conditions(n, 0).cuda()


def run(label):
    out = m.sample(cond, n, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.sample(cond, n, seed=2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ps = n * steps / dt
    print(f"{label:44s} {dt * 1e3:9.1f} ms  {ps / 1e6:7.2f} M patient-steps/s  (= {ps / 1000:8.0f} patients/s at T = 1000)  "
          f"{ps * 5193728 / 1e12:6.1f} TFLOP/s  [{m.last_sampler}/{m.last_precision}] max|x| {float(out.abs().max()):.3f}", flush=True)
    return out


ref = None
if do_fp32:
    m.precision = None
    m.sampler = "auto"
    ref = run("fp32 auto")
    m.sampler = "graph"
    run("fp32 per-layer (graph)")
m.precision = "bf16x3"
m.sampler = "auto"
for c in chunks:
    for s in streams:
        m.sample_chunk_rows, m.sample_streams = c, s
        out = run(f"bf16x3 chunk_rows {c} streams {s}")
if ref is not None:
    print("max |bf16x3 - fp32| / max|fp32| =", float((out - ref).abs().max() / ref.abs().max()))
