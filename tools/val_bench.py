#!/usr/bin/env python3
"""Throughput of the on-device validation metrics (BASELINE config 5 pieces) on synthetic data."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from osteosarcoma_diffusionmodel_amd.validation import BiologicalValidator
val = BiologicalValidator({"evaluation": {}})
g = torch.Generator(device="cuda").manual_seed(0)
D = 2000
for n in (20000, 50000):
    X = torch.randn(n, D, device="cuda", generator=g); Y = torch.randn(n, D, device="cuda", generator=g) * 1.02
    val.compute_mmd(X[:1024], Y[:1024])
    t0 = time.perf_counter(); mmd = val.compute_mmd(X, Y); dt = time.perf_counter() - t0
    print(f"MMD n=m={n} D={D}: {dt*1e3:.1f} ms, {3*2.0*n*n*D/dt/1e12:.1f} TFLOP/s (3 Gram blocks), mmd={mmd:.5f}")
for n in (100000, 1000000):
    X = torch.randn(n, 128, device="cuda", generator=g); Y = torch.randn(n, 128, device="cuda", generator=g)
    val.ks_tests(X[:1000], Y[:1000])
    t0 = time.perf_counter(); d, p = val.ks_tests(X, Y); dt = time.perf_counter() - t0
    print(f"KS 100 features n1=n2={n}: {dt*1e3:.1f} ms ({2*n*100/dt/1e9:.2f} G samples/s incl. host p-values), mean p={p.mean():.3f}")
    t0 = time.perf_counter(); c = val._mean_offdiag(X, list(range(0, 128, 2))); dt = time.perf_counter() - t0
    print(f"pathway coherence 64 genes n={n}: {dt*1e3:.2f} ms, mean corr={c:.2e}")
