#!/usr/bin/env python3
"""Headline benchmark: synthetic patients/sec for a full T=1000 reverse sample (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete T-step reverse sample (models/diffusion.py:427-449 incl. the mutation
threshold of utils/generate.py:135) of `--patients` conditional patients per GPU at the BASELINE
shape (D = 50 + 1900 + 50, hidden [256,512,256], cond 3, cosine schedule, three scenario
condition rows of config.yaml:124-141), on synthetic inputs already resident in HBM.  Rows are
sharded over ranks with no collective ("weak" scaling: per-GPU patients fixed).  Rank 0 prints ONE
JSON line; `roofline` is for the dominant fused GEMM kernel, `cpu_baseline` is the CPU oracle
(PyTorch CPU ops, validated against the reference) timed on this host on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
FLOP_PER_PATIENT_STEP = 5_193_728      # 2 x 2 596 864 MAC, GEMMs only

CONF = {"model": {"latent_dim": 128, "hidden_dims": [256, 512, 256], "gnn": {"dropout": 0.2},
                  "diffusion": {"num_steps": 1000, "beta_schedule": "cosine"},
                  "condition_on": ["survival_time", "event_occurred", "metastasis_at_diagnosis"]}}
SCENARIOS = [dict(survival_time=2000, event_occurred=0, metastasis_at_diagnosis=0),
             dict(survival_time=300, event_occurred=1, metastasis_at_diagnosis=1),
             dict(survival_time=800, event_occurred=0, metastasis_at_diagnosis=0)]


def scenario_conditions(n, offset):
    rows = torch.tensor([[(s["survival_time"] - 800) / 500, s["event_occurred"], s["metastasis_at_diagnosis"]]
                         for s in SCENARIOS], dtype=torch.float32)
    idx = (torch.arange(n) + offset) % 3
    return rows[idx]


def cpu_baseline(state_dict, budget_s=20.0):
    """CPU oracle p_sample steps at B=1024 on all host cores, extrapolated to T=1000."""
    from oracle import diffusion_oracle as O
    sd = {k: v.detach().cpu() for k, v in state_dict.items() if k.startswith(("condition_embed", "unet"))}
    bufs = O.schedule_buffers("cosine", 1000)
    B = 1024
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 2000, generator=g)
    cond = scenario_conditions(B, 0)
    z = torch.randn(B, 2000, generator=g)
    times = []
    t0 = time.perf_counter()
    with torch.no_grad():
        O.p_sample(sd, bufs, x, 500, cond, z, 3, 128)            # warm-up
        while time.perf_counter() - t0 < budget_s and len(times) < 40:
            s = time.perf_counter()
            O.p_sample(sd, bufs, x, 500, cond, z, 3, 128)
            times.append(time.perf_counter() - s)
    med = float(np.median(times))
    return {"value": B / (med * 1000), "unit": "patients/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"median of {len(times)} oracle p_sample steps at B={B}, D=2000 (x1000 steps extrapolated)",
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--patients", type=int, default=100_000, help="patients per GPU per step")
    ap.add_argument("--chunk-rows", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-rows", type=int, default=0, help="rows for the per-kernel event timing (default: chunk)")
    ap.add_argument("--profile-only", action="store_true", help="run only the per-kernel roofline leg (for rocprofv3 --stats)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # rehearsal knobs (single-GPU box): OSD_BENCH_BACKEND=gloo, OSD_BENCH_ONE_DEVICE=1 put every rank on cuda:0
    backend = os.environ.get("OSD_BENCH_BACKEND", "nccl")
    if os.environ.get("OSD_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, _lib as L
    torch.manual_seed(0)                      # default nn.Linear / GroupNorm init, as BASELINE.md section 3
    model = BiologyAwareDiffusionModel(50, 1900, 50, 3, CONF).to(dev).eval()
    if args.chunk_rows:
        model.sample_chunk_rows = args.chunk_rows
    if args.streams:
        model.sample_streams = args.streams
    model.use_graph = not args.no_graph

    n = args.patients
    offset = rank * n                          # global row ids: results independent of the GPU count
    cond = scenario_conditions(n, offset).to(dev)

    def step(i):
        out, mask = model.sample(cond, n, seed=1234 + i, row_offset=offset, return_mutation_mask=True)
        return out, mask

    if args.profile_only:
        args.steps, args.warmup, n = 0, 0, min(n, 1024)
        cond = cond[:n].contiguous()
    for i in range(args.warmup):
        step(i)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out, mask = step(args.warmup + i)
    fence()
    elapsed = max(time.perf_counter() - t0, 1e-9)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if args.steps:
        assert torch.isfinite(out).all().item()
    total_patients = n * world * args.steps
    value = total_patients / elapsed

    roof, cpu = None, None
    if rank == 0:
        # per-kernel durations with HIP events on the launch stream, same process, same shapes
        eng = model._engine()
        # rows per launch of the timed region: equal chunks of at most chunk_rows, whole 128-row tiles
        chunk_cap = model.sample_chunk_rows or 65536
        n_chunks = -(-args.patients // chunk_cap)
        chunk_rows = min(args.patients, (-(-args.patients // n_chunks) + 127) // 128 * 128)
        rows = args.profile_rows or chunk_rows
        if rows > cond.shape[0]:
            cond = scenario_conditions(rows, offset).to(dev)
        ms = (C.c_float * 64)()
        fl = (C.c_double * 64)()
        ne = C.c_int()
        pc = cond[:rows].contiguous()
        L.check(L.lib().osd_profile_step(eng.handle, L.ptr(pc), rows, 20, ms, fl, 64, C.byref(ne)))
        ne = ne.value
        names = ["input_proj"] + [f"block{i // 2}.{'first' if i % 2 == 0 else 'second'}" for i in range(ne - 2)] + ["output_proj+posterior"]
        launches = [{"launch": names[i], "ms": round(ms[i], 4), "tflops": round(fl[i] / (ms[i] * 1e-3) / 1e12, 2)} for i in range(ne)]
        # dominant kernel class = Linear+GroupNorm(groups of 64)+SiLU (the four 512-wide launches)
        wide = [i for i in range(1, ne - 1) if CONF["model"]["hidden_dims"][1] == 512 and launches[i]["launch"] in
                ("block0.first", "block0.second", "block3.first", "block3.second")]
        dom_ms = float(np.mean([ms[i] for i in wide]))
        dom_fl = float(np.mean([fl[i] for i in wide]))
        achieved = dom_fl / (dom_ms * 1e-3) / 1e12
        step_ms = float(sum(ms[i] for i in range(ne)))
        # HBM-side bytes per launch of that kernel from the rocprofv3 PMC passes (profiles/r01_pmc.md:
        # 2 x FETCH_SIZE + WRITE_SIZE at 32 768 rows), scaled to this launch's rows
        traffic = None
        tj = ROOT / "profiles" / "r01_traffic.json"
        if tj.exists():
            t = json.loads(tj.read_text())
            traffic = round(t["traffic_bytes_per_launch"]["GnSilu<64> glds"] * rows / t["rows_per_launch"])
        roof = {"bound": "mfma", "kernel": "gemm_glds_kernel<Tile<128,128,64,64>, EpiGnSilu<64>> (Linear+GroupNorm+SiLU, 512-wide layers)",
                "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_gbps": None if traffic is None else round(traffic / (dom_ms * 1e-3) / 1e9, 1),
                "avg_launch_ms": round(dom_ms, 4), "rows_per_launch": rows,
                "whole_step": {"ms": round(step_ms, 3),
                               "tflops": round(rows * FLOP_PER_PATIENT_STEP / (step_ms * 1e-3) / 1e12, 2)},
                "launches": launches}
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (one host, one CPU timing)
            cpu = cpu_baseline(model.state_dict())

    if rank == 0:
        line = {
            "metric": "synthetic patients/sec (full T-step reverse sample)", "value": round(value, 2), "unit": "patients/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / max(args.steps, 1), 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "reverse sampling of conditional patients, T=1000, D=2000 (50 mut + 1900 expr + 50 pathway), "
                                   "hidden [256,512,256], 3 scenario conditions, hipGraph-captured p_sample step",
                       "patients_per_gpu": n, "global_patients_per_step": n * world, "T": 1000,
                       "parallelism": f"patients sharded over {world} GPU(s), no collective",
                       "chunk_rows": model.sample_chunk_rows or 65536, "streams": model.sample_streams or 2,
                       "graph": model.use_graph},
            "achieved_tflops_end_to_end": round(value * 1000 * FLOP_PER_PATIENT_STEP / 1e12 / world, 2),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
