"""GPU: row-sharded validation (BASELINE config 5 / SURVEY section 8e): two ranks -- on cuda:0 with gloo carrying the
exchanges (runs on a 1-GPU box), and on one device each over nccl = RCCL (skipped below 2 GPUs) -- each hold half of the
synthetic patients; ``validate_all`` must return what one
process returns on all rows.  Exact for the integer statistics (KS, co-occurrence), 1e-6 for the fp32 Gram sums."""
import os
import socket

import numpy as np
import pandas as pd
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

EVAL = {"evaluation": {"driver_genes": ["TP53", "RB1"], "mutually_exclusive_pairs": [["TP53", "MDM2"]],
                       "required_correlations": [{"mutation": "TP53", "pathway": "PW0", "direction": "negative"},
                                                 {"mutation": "MDM2", "pathway": "PW1", "direction": "positive"}]}}


def _frames():
    rs = np.random.RandomState(8)
    names = ["TP53", "RB1", "MDM2"] + [f"M{i}" for i in range(17)]
    genes = [f"G{i}" for i in range(30)]
    n_real, n_syn = 80, 301
    pr = rs.rand(20) * 0.5 + 0.2
    rm = pd.DataFrame((rs.rand(n_real, 20) < pr).astype(np.float32), columns=names)
    sm = pd.DataFrame((rs.rand(n_syn, 20) < pr).astype(np.float32), columns=names)
    load = rs.randn(4, 30) * (rs.rand(4, 30) < 0.4)
    re = pd.DataFrame((rs.randn(n_real, 4) @ load + 0.6 * rs.randn(n_real, 30)).astype(np.float32), columns=genes)
    se = pd.DataFrame((rs.randn(n_syn, 4) @ load + 0.8 * rs.randn(n_syn, 30)).astype(np.float32), columns=genes)
    rp = pd.DataFrame(rs.randn(n_real, 2).astype(np.float32), columns=["PW0", "PW1"])
    sp = pd.DataFrame((rs.randn(n_syn, 2) + 0.5 * sm[["TP53", "MDM2"]].values).astype(np.float32), columns=["PW0", "PW1"])
    pgm = pd.DataFrame((rs.rand(30, 6) < 0.3).astype(int), index=genes, columns=[f"P{i}" for i in range(6)])
    return rm, re, rp, sm, se, sp, pgm


def _worker(rank, world, port, q, backend):
    import torch.distributed as dist
    from osteosarcoma_diffusionmodel_amd.parallel import shard_rows
    from osteosarcoma_diffusionmodel_amd.validation import BiologicalValidator
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rm, re, rp, sm, se, sp, pgm = _frames()
        off, cnt = shard_rows(len(sm), rank, world)
        sl = slice(off, off + cnt)
        val = BiologicalValidator(EVAL, sharded=True)
        assert val.comm.on
        np.random.seed(100 + rank)                   # ranks disagree on purpose: rank 0's draw is broadcast
        res = val.validate_all(rm, re, rp, sm.iloc[sl], se.iloc[sl], sp.iloc[sl], pgm)
        q.put((rank, res))
    except Exception as e:
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_sharded_validate_all_equals_single_process(backend):
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank")
    from osteosarcoma_diffusionmodel_amd.validation import BiologicalValidator
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    assert all(isinstance(r[1], dict) for r in res), res
    np.random.seed(100)                              # rank 0's seed
    ref = BiologicalValidator(EVAL).validate_all(*_frames())
    exact = {"mutation_frequency_correlation", "driver_gene_frequency_diff", "mutual_exclusivity_violation_rate",
             "cooccurrence_pattern_correlation", "ks_test_mean_pvalue", "ks_test_fraction_significant", "mutation_expression_violation_rate"}
    for _, r in res:
        assert set(r) == set(ref)
        for k, v in ref.items():
            tol = 1e-12 if k in exact else 2e-6
            assert abs(r[k] - v) <= tol * max(1.0, abs(v)), (k, r[k], v)
    assert res[0][1] == res[1][1]                    # identical numbers on every rank
