#!/usr/bin/env python3
"""Headline benchmark: synthetic patients/sec for a full T=1000 reverse sample (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete T-step reverse sample (models/diffusion.py:427-449 incl. the mutation
threshold of utils/generate.py:135) of `--patients` conditional patients per GPU at the BASELINE
shape (D = 50 + 1900 + 50, hidden [256,512,256], cond 3, cosine schedule, three scenario
condition rows of config.yaml:124-141), on synthetic inputs already resident in HBM.  Rows are
sharded over ranks with no collective ("weak" scaling: per-GPU patients fixed).  Rank 0 prints ONE
JSON line:
  * `roofline`      the dominant kernel of the timed region, measured live with HIP events;
  * `train`         the secondary metric (BASELINE config 2 at N = 1, config 4 = data-parallel at N > 1):
                    utils/train.py:204-250 per step (mixup, fused forward/backward, bucketed gradient
                    all-reduce over RCCL overlapped with backward, clip + AdamW) at 4096 rows per GPU;
  * `cpu_baseline`  the CPU oracle (PyTorch CPU ops, validated against the reference) timed on this
                    host: p_sample at B = 1024 and 4096, one un-extrapolated sample(N = 1024, T = 1000) and
                    one training step (N = 1 only).

Launched with WORLD_SIZE unset and --gpus N > 1 the script starts the N ranks itself (a child
`python -m torch.distributed.run`, started BEFORE this process touches the GPU) and returns the child's
exit code; under torchrun it is one of the ranks.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0         # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (at 2.4 GHz; the part holds ~1.75 GHz under this load)
HBM_PEAK_GBPS = 8000.0
FLOP_PER_PATIENT_STEP = 5_193_728      # 2 x 2 596 864 MAC, GEMMs only
TRAIN_FLOP_PER_SAMPLE = 14.56e6        # fwd 5.194 + dgrad 4.170 + wgrad 5.194 MFLOP
TRAIN_BATCH = 4096                     # per GPU (BASELINE configs 2 and 4)

CONF = {"model": {"latent_dim": 128, "hidden_dims": [256, 512, 256], "gnn": {"dropout": 0.2},
                  "diffusion": {"num_steps": 1000, "beta_schedule": "cosine"},
                  "condition_on": ["survival_time", "event_occurred", "metastasis_at_diagnosis"]}}
SCENARIOS = [dict(survival_time=2000, event_occurred=0, metastasis_at_diagnosis=0),
             dict(survival_time=300, event_occurred=1, metastasis_at_diagnosis=1),
             dict(survival_time=800, event_occurred=0, metastasis_at_diagnosis=0)]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--patients", type=int, default=100_000, help="patients per GPU per step")
    ap.add_argument("--chunk-rows", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--sampler", choices=["auto", "chain", "graph"], default="auto",
                    help="reverse-chain engine: persistent chain kernel, per-layer kernels under a hipGraph, or the library default")
    ap.add_argument("--chain-stagger", type=int, default=-1, help="chain kernel: cycles between the two workgroups of a CU (default: library)")
    ap.add_argument("--chain-steps-per-launch", type=int, default=-1)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-full-sample", action="store_true", help="skip the un-extrapolated CPU sample(N=1024, T=1000) leg")
    ap.add_argument("--no-train", action="store_true", help="skip the training leg")
    ap.add_argument("--train-steps", type=int, default=40)
    ap.add_argument("--profile-rows", type=int, default=0, help="rows for the per-kernel event timing (default: chunk)")
    ap.add_argument("--profile-only", action="store_true", help="run only the per-kernel roofline leg (for rocprofv3 --stats)")
    ap.add_argument("--train-only", action="store_true", help="run only the training leg (for rocprofv3 --stats)")
    ap.add_argument("--no-reference-workload", action="store_true", help="skip the reference's default generation workload leg")
    ap.add_argument("--reference-workload-only", action="store_true")
    ap.add_argument("--train-precision", default=None, choices=[None, "fp32", "bf16x3"],
                    help="diagnostic (profiles): run the `train` leg itself under this precision instead of fp32 + a split_bf16 sub-leg")
    ap.add_argument("--no-split", action="store_true", help="skip the opt-in bf16x3 split-precision leg (roofline.split_bf16): the same workload with every GEMM "
                    "on the bf16 matrix pipe at fp32 accuracy; never the headline")
    ap.add_argument("--no-mid-size", action="store_true", help="skip the 32 768-patient comparison (LDS-resident chain kernel vs per-layer kernels) inside the roofline object")
    ap.add_argument("--no-validate", action="store_true", help="skip the config-5 share leg (3 x 125 000 patients + on-device validation)")
    ap.add_argument("--validate-only", action="store_true")
    ap.add_argument("--validate-patients", type=int, default=125_000)
    return ap.parse_args()


def self_launch(args):
    """--gpus N without a torchrun environment: become the launcher.  Nothing in this process has touched the GPU
    (no HIP call, no torch.cuda.is_available()), and the ranks are CHILD processes, never an exec of this one."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def scenario_conditions(n, offset):
    import torch
    rows = torch.tensor([[(s["survival_time"] - 800) / 500, s["event_occurred"], s["metastasis_at_diagnosis"]]
                         for s in SCENARIOS], dtype=torch.float32)
    idx = (torch.arange(n) + offset) % 3
    return rows[idx]


# ------------------------------------------------------------------------------------------------
# CPU oracle legs (SURVEY section 8d "CPU baseline timing"); rank 0, N = 1 only
# ------------------------------------------------------------------------------------------------
def cpu_baseline(state_dict, full_sample=True, budget_s=12.0):
    import numpy as np
    import torch
    from oracle import diffusion_oracle as O
    sd = {k: v.detach().cpu() for k, v in state_dict.items() if k.startswith(("condition_embed", "unet"))}
    bufs = O.schedule_buffers("cosine", 1000)
    g = torch.Generator().manual_seed(1234)
    legs = {}

    def p_sample_leg(B, max_steps):
        x = torch.randn(B, 2000, generator=g)
        cond = scenario_conditions(B, 0)
        z = torch.randn(B, 2000, generator=g)
        times = []
        t0 = time.perf_counter()
        with torch.no_grad():
            O.p_sample(sd, bufs, x, 500, cond, z, 3, 128)            # warm-up
            while time.perf_counter() - t0 < budget_s and len(times) < max_steps:
                s = time.perf_counter()
                O.p_sample(sd, bufs, x, 500, cond, z, 3, 128)
                times.append(time.perf_counter() - s)
        med = float(np.median(times))
        return {"patients_per_s": B / (med * 1000), "ms_per_step": 1e3 * med, "steps_timed": len(times),
                "how": f"median oracle p_sample step at B={B}, D=2000, x1000 steps extrapolated"}

    legs["p_sample_b1024"] = p_sample_leg(1024, 40)
    legs["p_sample_b4096"] = p_sample_leg(4096, 12)
    # one training step: forward, autograd backward, clip_grad_norm_(1.0), AdamW -- utils/train.py:236-244
    B = TRAIN_BATCH
    x0 = torch.randn(B, 2000, generator=g)
    x0[:, :50] = (x0[:, :50] > 0).float()
    cond = torch.randn(B, 3, generator=g)
    masks = [(torch.rand(B, w, generator=g) >= 0.2).float() for w in (512, 256, 256, 512, 256)]
    names = list(sd)
    params = [sd[k].clone() for k in names]
    m1 = [torch.zeros_like(p) for p in params]
    m2 = [torch.zeros_like(p) for p in params]
    times = []
    for step in range(1, 4):
        t = torch.randint(0, 1000, (B,), generator=g)
        nz = torch.randn(B, 2000, generator=g)
        s = time.perf_counter()
        _, grads = O.training_loss_and_grads(dict(zip(names, params)), bufs, x0, cond, t, nz, 3, 128, masks, 0.2)
        gl, _ = O.clip_grad_norm([grads[k] for k in names], 1.0)
        O.adamw_step(params, gl, m1, m2, step, lr=1e-4, weight_decay=1e-5)
        times.append(time.perf_counter() - s)
    tmed = float(np.median(times[1:]))
    legs["train_step_b4096"] = {"samples_per_s": B / tmed, "ms_per_step": 1e3 * tmed, "steps_timed": len(times) - 1,
                                "how": "oracle forward + autograd backward + clip_grad_norm_ + AdamW at B=4096, D=2000, dropout 0.2"}
    value = legs["p_sample_b1024"]["patients_per_s"]
    sample = "median oracle p_sample step at B=1024, D=2000 (x1000 steps extrapolated)"
    # un-extrapolated: a whole sample(N = 1024, T = 1000) -- models/diffusion.py:427-449
    est = legs["p_sample_b1024"]["ms_per_step"]          # s for 1000 steps
    if full_sample and est <= 200.0:
        N = 1024
        cond = scenario_conditions(N, 0)
        x_T = torch.randn(N, 2000, generator=g)
        zg = torch.Generator().manual_seed(99)
        s = time.perf_counter()
        with torch.no_grad():
            out = O.sample(sd, bufs, cond, x_T, lambda t: torch.randn(N, 2000, generator=zg), 3, 128)
        dt = time.perf_counter() - s
        assert torch.isfinite(out).all().item()
        legs["full_sample_n1024"] = {"patients_per_s": N / dt, "seconds": dt, "how": "one complete oracle sample(N=1024, T=1000), not extrapolated"}
        value = N / dt
        sample = "one complete oracle sample(N=1024, T=1000, D=2000) incl. the normal draws, not extrapolated"
    elif full_sample:
        legs["full_sample_n1024"] = {"skipped": f"projected {est:.0f} s exceeds the 200 s budget of this leg"}
    return {"value": value, "unit": "patients/s", "cores": torch.get_num_threads(), "kind": "port", "sample": sample,
            "host_cpus": os.cpu_count(), "legs": legs}


# ------------------------------------------------------------------------------------------------
# the reference's own default generation workload (config/config.yaml:119-141, main.py:227-230, QUICKSTART.md:202)
# ------------------------------------------------------------------------------------------------
def reference_workload_leg(dev):
    """SyntheticPatientGenerator.generate_scenarios over the three config.yaml scenarios, T = 1000, at the reference's real
    TARGET-OS dims (62 mutations + 5054 expression genes + 26 pathways = 5142 features, D % 4 == 2) and at the BASELINE dims
    (50 + 1900 + 50): 1000 patients per scenario (utils/generate.py:146-175) and main.py's default split of
    generation.num_synthetic_samples = 1000 over the scenarios (333 each).  Wall time of the whole call, i.e. including the
    device-to-host copy and the numpy split of utils/generate.py:127-135.  QUICKSTART.md:202 quotes "5 min" GPU / "10 min" CPU
    for 1000 patients at the real dims on unspecified hardware (~3.3 / ~1.7 patients/s)."""
    import torch
    from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, SyntheticPatientGenerator
    scen = [{"name": n, "conditions": c} for n, c in zip(("early_stage_good_prognosis", "metastatic_poor_prognosis", "typical_patient"), SCENARIOS)]
    out = {"quickstart_reference": {"patients_per_s_gpu": 3.3, "patients_per_s_cpu": 1.7,
                                    "source": "QUICKSTART.md:202 (1000 patients, dims 62/5054/26, unspecified hardware)"}, "runs": []}
    for dims in ((62, 5054, 26), (50, 1900, 50)):
        torch.manual_seed(0)
        model = BiologyAwareDiffusionModel(*dims, 3, CONF).to(dev).eval()
        gen = SyntheticPatientGenerator(model, CONF, device=dev)
        D = sum(dims)
        flop_row_step = 2 * (2 * D * 256 + 1_572_864)                # input_proj + trunk + output_proj, hidden [256, 512, 256]
        gen.generate(128, scen[0]["conditions"])                    # warm-up: kernel loading, workspace, graph capture
        torch.cuda.synchronize()
        for per in (1000, 333):
            t0 = time.perf_counter()
            res = gen.generate_scenarios(scen, per)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert all(r["expression"].shape == (per, dims[1]) for r in res.values())
            n = per * len(scen)
            out["runs"].append({"dims": list(dims), "D": D, "patients_per_scenario": per, "patients": n, "T": 1000,
                                "seconds": round(dt, 4), "patients_per_s": round(n / dt, 1),
                                "engine": model.last_sampler + ("/" + model.last_chain_variant if model.last_chain_variant else ""),
                                "achieved_tflops": round(n * 1000 * flop_row_step / dt / 1e12, 2),
                                "frac_of_fp32_mfma_peak": round(n * 1000 * flop_row_step / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                                "vs_quickstart_gpu": round(n / dt / 3.3, 1)})
        del model, gen
    return out


# ------------------------------------------------------------------------------------------------
# BASELINE config 5, one GPU's share: 3 scenarios x 125 000 patients + on-device validation (utils/validation.py:225-387)
# ------------------------------------------------------------------------------------------------
def validate_leg(model, dev, per_scenario=125_000):
    """Config 5 = 3 clinical scenarios x 1 000 000 patients over 8 GPUs; one GPU's share is 3 x 125 000.  Each scenario is
    generated (full T = 1000 chain) and validated against a 125 000-row "real" cohort (synthetic: an earlier sample with another
    seed) -- per metric the device time and the roofline that bounds it, then BiologicalValidator.validate_all through the
    reference's entry point (DeviceFrame inputs; its Wasserstein-on-PCA part is the reference's own host-side sklearn / scipy
    code).  MMD: three Gram blocks on fp32 MFMA; K_XX / K_YY on their upper triangles: 2 D (n(n+1)/2 + m(m+1)/2 + n m) necessary FLOP.  KS / pathway coherence / co-occurrence: HBM
    streams; algorithmic bytes = every element they must read once."""
    import numpy as np
    import pandas as pd
    import torch
    from osteosarcoma_diffusionmodel_amd.validation import BiologicalValidator, DeviceFrame
    n, D, md, ed = per_scenario, 2000, 50, 1900
    mut_cols = [f"GENE_{i}" for i in range(md)]
    expr_cols = [f"EXPR_{i}" for i in range(ed)]
    path_cols = [f"PATHWAY_{i}" for i in range(D - md - ed)]
    rules = [{"mutation": "GENE_0", "pathway": "PATHWAY_0", "direction": "positive"}, {"mutation": "GENE_1", "pathway": "PATHWAY_1", "direction": "negative"}]
    val = BiologicalValidator({"evaluation": {"driver_genes": mut_cols[:4], "mutually_exclusive_pairs": [[mut_cols[0], mut_cols[1]]],
                                              "required_correlations": rules}}, device=str(dev))
    # gene x pathway membership: 10 pathways of 64 expression genes each (pathway coherence uses the first 10 pathways, :144)
    pgm = pd.DataFrame(0, index=expr_cols, columns=[f"HALLMARK_{i}" for i in range(10)])
    for i in range(10):
        pgm.iloc[64 * i:64 * (i + 1), i] = 1

    def sample(seed, cond_row):
        cond = torch.tensor([cond_row], dtype=torch.float32, device=dev).repeat(n, 1)
        x, mask = model.sample(cond, n, seed=seed, return_mutation_mask=True)
        x = torch.nan_to_num(x, nan=0.0, posinf=1e6, neginf=-1e6)      # random-init weights: the un-clamped chain reaches 3e5
        x[:, :md] = mask
        return x

    def frames(x):
        return (DeviceFrame(x[:, :md].contiguous(), mut_cols), DeviceFrame(x[:, md:md + ed].contiguous(), expr_cols),
                DeviceFrame(x[:, md + ed:].contiguous(), path_cols))

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0

    rows = [[(s["survival_time"] - 800) / 500, s["event_occurred"], s["metastasis_at_diagnosis"]] for s in SCENARIOS]
    real = sample(777, rows[2])
    rm, re_, rp = frames(real)
    val.compute_mmd(real[:2048], real[:2048])            # warm-up (kernel loading)
    val.ks_tests(real[:2048], real[:2048])
    out = {"config": "BASELINE config 5, one GPU's share: 3 scenarios x %d patients, T = 1000, validated against %d real rows" % (n, n),
           "scenarios": []}
    for i, row in enumerate(rows):
        x, t_sample = timed(lambda: sample(2000 + i, row))
        sm, se, sp = frames(x)
        # device kernels on their own (no host-side statistics), then the whole of validate_all as the reference runs it
        mmd, t_mmd = timed(lambda: val.compute_mmd(real, x))
        _, t_ks = timed(lambda: val.k.ks_extremes(real, x, 100))
        cols64 = list(range(64))
        _, t_coh = timed(lambda: (val._mean_offdiag(re_.values, cols64), val._mean_offdiag(se.values, cols64)))
        _, t_gram = timed(lambda: (val._gram(sm.values, list(range(50))), val._column_sums(sm.values)))
        allr, t_all = timed(lambda: val.validate_all(rm, re_, rp, sm, se, sp, pathway_gene_matrix=pgm))
        # necessary work only: K_XX and K_YY are symmetric and run on the tiles on or above the diagonal (csrc/validate.hip)
        flop_mmd = 2.0 * D * (n * (n + 1.0) / 2 * 2 + n * float(n))
        ks_bytes = 100 * 2 * n * 4.0                         # every sample of the 100 tested features once
        coh_bytes = 2 * 64 * 2 * n * 4.0                     # real + synthetic, 64 genes, moments pass + row pass
        gram_bytes = 2 * 50 * n * 4.0
        out["scenarios"].append({
            "sample_s": round(t_sample, 3), "patients_per_s": round(n / t_sample, 1),
            "mmd": {"ms": round(1e3 * t_mmd, 2), "value": round(float(mmd), 6), "bound": "mfma", "tflops": round(flop_mmd / t_mmd / 1e12, 2),
                    "frac": round(flop_mmd / t_mmd / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)},
            "ks_extremes_100_features": {"ms": round(1e3 * t_ks, 2), "bound": "hbm", "algorithmic_gbps": round(ks_bytes / t_ks / 1e9, 1),
                                         "frac": round(ks_bytes / t_ks / 1e9 / HBM_PEAK_GBPS, 4),
                                         "note": "column gather + 4-pass segmented radix sort + binary-search extremes; p-values are host scalars (scipy kstwo.sf, as ks_2samp)"},
            "pathway_coherence_64_genes": {"ms": round(1e3 * t_coh, 3), "bound": "hbm", "algorithmic_gbps": round(coh_bytes / t_coh / 1e9, 1),
                                           "frac": round(coh_bytes / t_coh / 1e9 / HBM_PEAK_GBPS, 4)},
            "cooccurrence_gram_50_genes": {"ms": round(1e3 * t_gram, 3), "bound": "hbm", "algorithmic_gbps": round(gram_bytes / t_gram / 1e9, 1),
                                           "frac": round(gram_bytes / t_gram / 1e9 / HBM_PEAK_GBPS, 4)},
            "validate_all": {"s": round(t_all, 3), "overall_biological_score": round(float(allr.get("overall_biological_score", float("nan"))), 4),
                             "mmd": round(float(allr["mmd"]), 6), "ks_test_mean_pvalue": round(float(allr["ks_test_mean_pvalue"]), 4),
                             "keys": len(allr),
                             "note": "utils/validation.py:300-387 end to end: the device metrics + the reference's own host-side parts (100 scipy kstwo.sf "
                                     "p-values at N = 62 500, 2 x 1225 chi-square tables, PCA(10) + Wasserstein on 2 x %d x %d values)" % (n, D)}})
        del x, sm, se, sp
    return out


# ------------------------------------------------------------------------------------------------
# training leg (BASELINE config 2 / 4)
# ------------------------------------------------------------------------------------------------
def train_leg(dev, dist, world, rank, steps, backend, precision=None, epochs=True):
    """utils/train.py:204-250 per step on a device-resident synthetic dataset of 65 536 rows per rank (SURVEY section
    8d, config 2): mixup (host draws as the reference), osd_train_loss_fwd_bwd, bucketed all-reduce, fused clip+AdamW.
    Returns (per-rank dict).  `exposed_comm_ms` = time the handle's stream waits for the gradient exchange after its own
    backward has finished (event pair around the wait), averaged per step."""
    import torch
    from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel
    from osteosarcoma_diffusionmodel_amd.train import Trainer
    B = TRAIN_BATCH
    conf = {"model": dict(CONF["model"])}
    conf["training"] = {"learning_rate": 1e-4, "weight_decay": 1e-5, "patience": 100, "min_delta": 1e-4,
                        "augmentation": {"mixup_alpha": 0.2}, "save_dir": "/tmp/osd_bench_ckpt", "num_epochs": 1,
                        "save_frequency": 10, "val_split": 0.2, "random_seed": 42, "batch_size": B}
    torch.manual_seed(0)
    model = BiologyAwareDiffusionModel(50, 1900, 50, 3, conf)
    model.precision = precision        # "bf16x3": the weight gradients of the step on the bf16 matrix pipe (csrc/wgrad_group.h); None: fp32
    tr = Trainer(model, [], [], conf, device=dev)
    model.train()
    g = torch.Generator(device=dev).manual_seed(42 + rank)
    rows = 65536
    data = torch.randn(rows, 2000, device=dev, generator=g)
    data[:, :50] = (torch.rand(rows, 50, device=dev, generator=g) < 0.5).float()
    cond = torch.randn(rows, 3, device=dev, generator=g)
    surv = torch.rand(rows, device=dev, generator=g)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

    order = torch.arange(rows, device=dev, dtype=torch.int64)
    # the mixup draws are the reference's (np.random.beta, torch.randperm on the host), made up front and uploaded once, as
    # Trainer.train_epoch does per epoch (MixupAugmentation.draw_epoch)
    lams, perms, _ = tr.mixup.draw_epoch([B] * (20 + steps), dev)

    def one(i, timed):
        # a batch of the device-resident dataset (Trainer's epoch path, ResidentSplit): rows gathered, mixed and noised by one kernel
        o = (i * B) % (rows - B)
        idx = order[o:o + B]
        j = i if timed else steps + i
        lam, perm = lams[j], perms[j]
        # the exposed-communication event pair only exists under data parallel: two timing events cost the single-GPU step
        # two barrier packets (~10 us) for a number that is zero by construction
        return tr.train_step(None, None, source=(data, cond, surv, idx, idx[perm], lam), comm_events=ev[i] if (timed and world > 1) else None)

    # 20 untimed steps: the first few pay one-off costs (kernel loading, work-list uploads, allocator growth), and on every box
    # one more host-side stall of ~4 ms shows up between steps 10 and 15 (tools/probes/train_steps.py); from step 15 on the
    # step time is flat
    for i in range(20):
        one(i, False)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = one(i, True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    exposed = 0.0
    if world > 1:
        exposed = sum(a.elapsed_time(b) for a, b in ev) / steps
    per_step = dt / steps
    # the same steps through Trainer.train_epoch over loaders shaped like prepare_data's (utils/train.py:413-437: random_split,
    # shuffle, drop_last, batch 4096): device-resident epoch path vs the reference's per-batch host hand-over
    epoch = None
    if world == 1 and epochs:
        from osteosarcoma_diffusionmodel_amd.train import OsteosarcomaDataset
        ds = object.__new__(OsteosarcomaDataset)
        ds.data, ds.conditions, ds.survival_days = data.cpu(), cond.cpu(), surv.cpu()
        tr_ds, va_ds = torch.utils.data.random_split(ds, [rows - 8192, 8192], generator=torch.Generator().manual_seed(42))
        epoch = {}
        for resident in (True, False):
            tr.train_loader = torch.utils.data.DataLoader(tr_ds, batch_size=B, shuffle=True, num_workers=0, drop_last=True)
            tr.val_loader = torch.utils.data.DataLoader(va_ds, batch_size=B, shuffle=False, num_workers=0)
            tr.resident = None if resident else False
            n_ep = 4 if resident else 1
            tr.train_epoch()                                   # upload / warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_ep):
                tr.train_epoch()
            torch.cuda.synchronize()
            e_dt = (time.perf_counter() - t0) / n_ep
            n_b = len(tr.train_loader)
            epoch["resident" if resident else "dataloader"] = {"samples_per_s": round(n_b * B / e_dt, 1), "ms_per_step": round(1e3 * e_dt / n_b, 4),
                                                               "steps_per_epoch": n_b}
        epoch["resident_vs_train_step"] = round(epoch["resident"]["samples_per_s"] / (B / per_step), 3)
    tf = TRAIN_FLOP_PER_SAMPLE * B * world / per_step / 1e12
    return {"config": ("DDP diffusion training, %d x MI355X, global batch %d" % (world, B * world)) if world > 1 else
                      "diffusion training, D=2000, T=1000, batch 4096, 1 x MI355X",
            "ms_per_step": round(1e3 * per_step, 4), "samples_per_s": round(B * world / per_step, 1), "steps": steps,
            "global_batch": B * world, "per_gpu_batch": B, "achieved_tflops": round(tf, 2),
            "frac_of_fp32_mfma_peak": round(tf / world / FP32_MFMA_PEAK_TFLOPS, 4),
            "exposed_comm_ms_per_step": round(exposed, 4), "grad_message_bytes": int(tr.flat.grad.numel()) * 4,
            "comm": None if world == 1 else f"{tr.comm_kind} ({backend})", "final_loss": round(float(loss.item()), 5),
            "epoch_samples_per_s": None if epoch is None else epoch["resident"]["samples_per_s"], "train_epoch": epoch,
            "includes": "mixup + q_sample + forward + backward + bucketed all-reduce + clip_grad_norm_ + AdamW, dropout 0.2 (Philox)"}


def train_split_leg(dev, steps):
    """The same training step with ``model.precision = "bf16x3"``: the grouped weight-gradient launch -- a quarter of the fp32 step,
    leaves of the graph -- on the bf16 matrix pipe at fp32 accuracy (csrc/wgrad_group.h: operands split as they are staged); forward
    and the dgrad chain stay fp32 (latency-bound launches of 10-50 us).  Opt-in, not the `train` figure."""
    r = train_leg(dev, None, 1, 0, steps, None, precision="bf16x3", epochs=False)
    return {"dtype": "weight gradients and output_proj + MSE: bf16x3 split, f32 accumulate; the rest of forward / dgrad: f32", "ms_per_step": r["ms_per_step"],
            "samples_per_s": r["samples_per_s"], "final_loss": r["final_loss"],
            "effective_tflops": r["achieved_tflops"], "effective_vs_fp32_mfma_peak": r["frac_of_fp32_mfma_peak"]}


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))

    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # rehearsal knob (single-GPU box): OSD_BENCH_ONE_DEVICE=1 puts every rank on cuda:0; RCCL refuses two ranks on one
    # device, so the exchange then runs over gloo (reported in comm_backend)
    one_device = os.environ.get("OSD_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("OSD_BENCH_BACKEND", "gloo" if one_device else "nccl")
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    comm_ranks = 1
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        ones = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        comm_ranks = int(ones.item())
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, _lib as L
    torch.manual_seed(0)                      # default nn.Linear / GroupNorm init, as BASELINE.md section 3
    model = BiologyAwareDiffusionModel(50, 1900, 50, 3, CONF).to(dev).eval()
    if args.chunk_rows:
        model.sample_chunk_rows = args.chunk_rows
    if args.streams:
        model.sample_streams = args.streams
    model.use_graph = not args.no_graph
    if args.sampler != "auto":
        model.sampler = args.sampler
    if args.chain_stagger >= 0:
        model.chain_stagger = args.chain_stagger
    if args.chain_steps_per_launch >= 0:
        model.chain_steps_per_launch = args.chain_steps_per_launch

    n = args.patients
    offset = rank * n                          # global row ids: results independent of the GPU count
    cond = scenario_conditions(n, offset).to(dev)

    def step(i):
        out, mask = model.sample(cond, n, seed=1234 + i, row_offset=offset, return_mutation_mask=True)
        return out, mask

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    line = {}
    if args.reference_workload_only:
        if rank == 0:
            print(json.dumps({"reference_workload": reference_workload_leg(dev)}), flush=True)
        return
    if args.validate_only:
        if rank == 0:
            print(json.dumps({"validate": validate_leg(model, dev, args.validate_patients)}), flush=True)
        return
    if args.train_only:
        tr = train_leg(dev, dist, world, rank, args.train_steps, backend, precision=args.train_precision)
        if world == 1 and not args.no_split and not args.train_precision:
            tr["split_bf16"] = train_split_leg(dev, args.train_steps)
        if rank == 0:
            print(json.dumps({"train": tr}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.profile_only:
        args.steps, args.warmup, n = 0, 0, min(n, 1024)
        cond = cond[:n].contiguous()
    for i in range(args.warmup):
        step(i)
    fence()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        evs[i][0].record()                     # current stream = the stream the chain kernel is launched on
        out, mask = step(args.warmup + i)
        evs[i][1].record()
    fence()
    elapsed = max(time.perf_counter() - t0, 1e-9)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if args.steps:
        if not os.environ.get("OSD_BENCH_TIMING_ONLY"):     # timing experiments with garbage-producing diagnostic builds
            assert torch.isfinite(out).all().item()
        del out, mask
    total_patients = n * world * args.steps
    value = total_patients / elapsed
    engine_used = model.last_sampler if hasattr(model, "last_sampler") else "graph"

    roof, cpu = None, None
    sample_ms = [a.elapsed_time(b) for a, b in evs] if args.steps else []
    if rank == 0:
        roof = roofline_leg(model, args, cond, offset, dev, engine_used, sample_ms, step)
    train = None
    if not args.no_train and not args.profile_only:
        train = train_leg(dev, dist, world, rank, args.train_steps, backend)
        if world == 1 and not args.no_split:
            train["split_bf16"] = train_split_leg(dev, args.train_steps)
    refw = None
    if rank == 0 and world == 1 and not args.no_reference_workload and not args.profile_only:
        refw = reference_workload_leg(dev)
    vleg = None
    if rank == 0 and world == 1 and not args.no_validate and not args.profile_only:
        vleg = validate_leg(model, dev, args.validate_patients)
    if rank == 0 and not args.no_cpu_baseline and world == 1 and not args.profile_only:
        cpu = cpu_baseline(model.state_dict(), full_sample=not args.no_cpu_full_sample)   # N = 1 only (one host, one CPU timing)

    if rank == 0:
        line = {
            "metric": "synthetic patients/sec (full T-step reverse sample)", "value": round(value, 2), "unit": "patients/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / max(args.steps, 1), 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "reverse sampling of conditional patients, T=1000, D=2000 (50 mut + 1900 expr + 50 pathway), "
                                   "hidden [256,512,256], 3 scenario conditions (BASELINE config 3)",
                       "patients_per_gpu": n, "global_patients_per_step": n * world, "T": 1000,
                       "parallelism": f"patients sharded over {world} GPU(s), no collective",
                       "sampler": engine_used, "chunk_rows": model.sample_chunk_rows or 65536, "streams": model.sample_streams or 2,
                       "graph": model.use_graph},
            "comm_ranks": comm_ranks, "comm_backend": None if world == 1 else backend,
            "rccl_ranks": comm_ranks if (world > 1 and backend == "nccl") else None,       # only when the all-reduce really ran over RCCL
            "achieved_tflops_end_to_end": round(value * 1000 * FLOP_PER_PATIENT_STEP / 1e12 / world, 2),
            "frac_of_fp32_mfma_peak_end_to_end": round(value * 1000 * FLOP_PER_PATIENT_STEP / 1e12 / world / FP32_MFMA_PEAK_TFLOPS, 4),
            "roofline": roof, "train": train, "reference_workload": refw, "validate": vleg, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def split_leg(model, args, cond, offset, step_fn):
    """The SAME workload (args.patients, full T, same seeds -> same Philox draws) under ``model.precision = "bf16x3"`` (csrc/gemm_bf3.h,
    split.hip): every fp32 operand as three bf16 planes (exact), six v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation -- the
    fp32 tolerances of the parity tests unchanged (tests/test_gpu_split.py).  Opt-in, never the headline: `value`, `dtype` and the
    default engine of this line are the fp32 ones.  `executed_mfma_tflops` = 6 x the algorithmic FLOP, priced against the bf16 peak;
    `effective_tflops` = the algorithmic FLOP (what the fp32 kernels are priced on), shown against the fp32 MFMA peak it is not bound by."""
    import torch
    if args.no_split or args.profile_only or int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
        return None
    T = CONF["model"]["diffusion"]["num_steps"]
    keep_p, keep_s = model.precision, model.sampler
    ref, _ = step_fn(20_000)
    try:
        model.precision = "bf16x3"
        model.sampler = "auto"
        model.sample(cond[:4096].contiguous(), 4096, seed=5)          # weight planes, workspace, graph capture off the clock
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, _ = step_fn(20_000)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert model.last_precision == "bf16x3"
    finally:
        model.precision, model.sampler = keep_p, keep_s
    flop = float(args.patients) * T * FLOP_PER_PATIENT_STEP
    scale = float(ref.abs().max())
    return {"dtype": "bf16x3 split, f32 accumulate", "patients_per_s": round(args.patients / dt, 1), "seconds": round(dt, 4),
            "engine": "per-layer launches on planes buffers (split.hip), hipGraph replay",
            "effective_tflops": round(flop / dt / 1e12, 2), "effective_vs_fp32_mfma_peak": round(flop / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "executed_mfma_tflops": round(6 * flop / dt / 1e12, 1), "bf16_mfma_peak": BF16_MFMA_PEAK_TFLOPS,
            "frac_of_bf16_mfma_peak": round(6 * flop / dt / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
            "max_abs_diff_vs_fp32_engine_over_max_abs": float((out - ref).abs().max()) / max(scale, 1e-30),
            "note": "same seeds as the fp32 run (identical Philox draws); full T-step chains of random-init weights amplify rounding differences, "
                    "the parity statement is tests/test_gpu_split.py (oracle, unchanged tolerances)"}


def roofline_leg(model, args, cond, offset, dev, engine_used, sample_ms, step_fn):
    """The dominant kernel of the timed region, measured live with HIP events on its launch stream.

    chain engine: ONE kernel (chain_kernel) runs the whole T-step reverse sample of the rank's patients, so its launch
    duration is the event pair around sample() in the timed region (the conditioning GEMMs and the x_T fill that precede it
    on the same stream are < 0.1 % of it) and its algorithmic work is patients x T x 5 193 728 FLOP.
    per-layer engine: 12 launches per step; per-launch durations from osd_profile_step (eager launches on the handle's
    stream), the dominant class being the 512-wide Linear+GroupNorm+SiLU launches."""
    import numpy as np
    import torch
    from osteosarcoma_diffusionmodel_amd import _lib as L
    eng = model._engine()
    T = CONF["model"]["diffusion"]["num_steps"]
    traffic_files = [ROOT / "profiles" / n for n in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json")]
    import hashlib
    lib_sha = hashlib.sha256(Path(os.environ.get("OSDIFF_LIB", L.LIB_PATH)).read_bytes()).hexdigest()

    def traffic_for(key, scale_units):
        """HBM-side bytes per launch from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE) -- quoted only when the file was collected
        on THIS build of the library (its recorded sha256 equals the loaded .so's); a stale file gives traffic: null, not a figure."""
        for tj in traffic_files:
            if tj.exists():
                t = json.loads(tj.read_text())
                ent = t.get("traffic_bytes_per_launch", {}).get(key)
                if ent is None:
                    continue
                if t.get("library_sha256") != lib_sha:
                    return None, (f"profiles/{tj.name} was collected on another build of libosdiff.so (sha256 {str(t.get('library_sha256'))[:12]}... "
                                  f"vs loaded {lib_sha[:12]}...): no traffic figure quoted for this run; tools/round_pmc.sh regenerates it")
                per_unit = ent / t["units_per_launch"][key] if "units_per_launch" in t else ent / t["rows_per_launch"]
                return round(per_unit * scale_units), (f"profiles/{tj.name}: rocprofv3 --pmc passes of THIS build of the library (sha256 match) in a separate run "
                                                       f"(2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction), scaled by work; not measured inside the timed region")
        return None, None

    # ---- per-layer engine: per-launch table (also reported when the chain engine is the default: it is the fallback path) ----
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
    launches = [{"launch": names[i], "ms": round(ms[i], 4), "tflops": round(fl[i] / (ms[i] * 1e-3) / 1e12, 2),
                 "frac": round(fl[i] / (ms[i] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 3)} for i in range(ne)]
    wide = [i for i in range(1, ne - 1) if launches[i]["launch"] in ("block0.first", "block0.second", "block3.first", "block3.second")]
    dom_ms = float(np.mean([ms[i] for i in wide]))
    dom_fl = float(np.mean([fl[i] for i in wide]))
    layer_achieved = dom_fl / (dom_ms * 1e-3) / 1e12
    step_ms = float(sum(ms[i] for i in range(ne)))
    ltraffic, lsource = traffic_for("GnSilu<64> glds", rows)
    per_layer = {"kernel": "gemm_glds_kernel<Tile<128,128,64,64>, EpiGnSilu<64>> (Linear+GroupNorm+SiLU, 512-wide layers)",
                 "achieved": round(layer_achieved, 2), "frac": round(layer_achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                 "avg_launch_ms": round(dom_ms, 4), "rows_per_launch": rows, "traffic": ltraffic, "traffic_source": lsource,
                 "whole_step": {"ms": round(step_ms, 3), "tflops": round(rows * FLOP_PER_PATIENT_STEP / (step_ms * 1e-3) / 1e12, 2)},
                 "launches": launches}
    if engine_used != "chain" or not sample_ms:
        roof = {"bound": "mfma", "kernel": per_layer["kernel"], "achieved": per_layer["achieved"], "peak": FP32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": per_layer["frac"], "traffic": ltraffic, "traffic_source": lsource,
                "traffic_gbps": None if ltraffic is None else round(ltraffic / (dom_ms * 1e-3) / 1e9, 1),
                "avg_launch_ms": per_layer["avg_launch_ms"], "rows_per_launch": rows, "whole_step": per_layer["whole_step"],
                "launches": launches}
        return roof
    # ---- chain engine ----
    launch_ms = float(np.mean(sample_ms))
    flop = float(args.patients) * T * FLOP_PER_PATIENT_STEP
    achieved = flop / (launch_ms * 1e-3) / 1e12
    ctraffic, csource = traffic_for("chain_kernel", args.patients * T)
    # one step of the per-layer engine (hipGraph replay, two chunks in flight) on the same inputs, for comparison
    keep = model.sampler
    model.sampler = "graph"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_fn(10_000)
    torch.cuda.synchronize()
    g_s = time.perf_counter() - t0
    model.sampler = keep
    per_layer["patients_per_s_one_step"] = round(args.patients / g_s, 1)
    # the mid-size batch: 32 768 patients, full T -- below the workspace chain's 512-tile threshold auto runs the LDS-resident chain
    # kernel (csrc/chain_panel.h; 64 patients per workgroup, activations in LDS), beside the per-layer kernels on the same rows
    mid = None
    if not args.no_mid_size and not args.profile_only and int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1:
        mrows = 32768
        mcond = scenario_conditions(mrows, offset).to(dev)
        mid = {"rows": mrows, "steps": T}
        for name, sampler in (("auto", "auto"), ("per_layer", "graph")):
            model.sampler = sampler
            model.sample(mcond[:2048], 2048, seed=11)          # engine set-up off the clock
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.sample(mcond, mrows, seed=12)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            tf = mrows * T * FLOP_PER_PATIENT_STEP / dt / 1e12
            mid[name] = {"patients_per_s": round(mrows / dt, 1), "tflops": round(tf, 2), "frac": round(tf / FP32_MFMA_PEAK_TFLOPS, 4),
                         "engine": model.last_sampler + ("/" + model.last_chain_variant if model.last_chain_variant else "")}
        model.sampler = keep
    split = split_leg(model, args, cond, offset, step_fn)
    return {"split_bf16": split, "bound": "mfma", "kernel": "chain_kernel (persistent: all 12 layers x all T steps of the rank's patients in one launch; "
                                       "v_mfma_f32_32x32x2_f32, 128x128 tiles, LDS-DMA staging)",
            "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
            "traffic": ctraffic, "traffic_source": csource,
            "traffic_gbps": None if ctraffic is None else round(ctraffic / (launch_ms * 1e-3) / 1e9, 1),
            "avg_launch_ms": round(launch_ms, 2), "launches_timed": len(sample_ms),
            "algorithmic_flop_per_launch": flop, "patients_per_launch": args.patients, "steps_per_launch": T,
            "algorithmic_hbm_bytes_per_launch": float(args.patients) * T * 16000,
            "per_layer_engine": per_layer, "mid_size_batch": mid}


if __name__ == "__main__":
    main()
