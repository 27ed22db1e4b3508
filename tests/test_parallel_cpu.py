"""CPU, world_size 2 over gloo: the data-parallel host logic (osteosarcoma_diffusionmodel_amd/parallel.py)
that bench.py and Trainer use on N GPUs -- patient sharding with no collective, and the bucketed
gradient all-reduce in backward order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from osteosarcoma_diffusionmodel_amd import _lib as L
from osteosarcoma_diffusionmodel_amd.parallel import allreduce_buckets, bucket_slices, shard_rows


def test_shard_rows_partition():
    for n in (0, 1, 7, 100000, 3000000):
        for world in (1, 2, 3, 8):
            spans = [shard_rows(n, r, world) for r in range(world)]
            assert spans[0][0] == 0
            assert sum(c for _, c in spans) == n
            for (o0, c0), (o1, _) in zip(spans, spans[1:]):
                assert o0 + c0 == o1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def _cfg():
    import ctypes as C
    cfg = L.OsdConfig()
    cfg.mutation_dim, cfg.expression_dim, cfg.pathway_dim, cfg.condition_dim = 50, 1900, 50, 3
    cfg.time_dim, cfg.n_hidden = 128, 3
    for i, v in enumerate((256, 512, 256)):
        cfg.hidden_dims[i] = v
    cfg.num_steps, cfg.dropout_p = 1000, 0.2
    return cfg


def _buckets():
    import ctypes as C
    cfg = _cfg()
    lib = L.lib()
    nb = lib.osd_grad_buckets(C.byref(cfg), None, None, 0)
    first, last = (C.c_int32 * nb)(), (C.c_int32 * nb)()
    lib.osd_grad_buckets(C.byref(cfg), first, last, nb)
    numel = [lib.osd_param_numel(C.byref(cfg), i) for i in range(lib.osd_num_params(C.byref(cfg)))]
    return [(int(first[i]), int(last[i])) for i in range(nb)], numel


def test_grad_buckets_cover_every_parameter_once_in_backward_order():
    buckets, numel = _buckets()
    assert buckets[0] == (50, 51)                       # output_proj is final first
    assert buckets[-1] == (0, 9)                        # embeddings + input/cond/time proj last
    covered = sorted(i for f, l in buckets for i in range(f, l + 1))
    assert covered == list(range(52))
    offsets = np.concatenate([[0], np.cumsum(numel)])
    sl = bucket_slices(offsets, buckets)
    assert sum(e - s for s, e in sl) == 2663952
    assert all(s0 >= e1 for (s0, _), (_, e1) in zip(sl, sl[1:]))      # descending, non-overlapping


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        buckets, numel = _buckets()
        offsets = np.concatenate([[0], np.cumsum(numel)])
        sl = bucket_slices(offsets, buckets)
        g = torch.Generator().manual_seed(100 + rank)
        local = torch.randn(int(offsets[-1]), generator=g)          # this rank's gradient
        flat = local / world                                          # the loss_scale = 1/world fold
        allreduce_buckets(flat, sl)                                   # CPU path: no streams / events
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        mean = torch.stack(gathered).mean(0)
        ok = torch.allclose(flat, mean, rtol=1e-6, atol=1e-7)
        # sampling shards: offsets are disjoint and ordered, no communication needed
        off, cnt = shard_rows(100001, rank, world)
        spans = [None] * world
        dist.all_gather_object(spans, (off, cnt))
        ok = ok and sum(c for _, c in spans) == 100001 and spans[0][0] == 0
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_is_the_mean_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


# ---- row-sharded validation (third row): the combination logic over gloo, numpy standing in for
# ---- the device kernels (the GPU tests run the same functions with the real kernels) ---------------------------------
class NumpyKernels:
    """Per-shard partial results with the semantics of validation.DeviceKernels, float64 numpy."""

    @staticmethod
    def rbf_sum(a, b, gamma):
        a, b = a.double().numpy(), b.double().numpy()
        d2 = (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * a @ b.T
        return float(np.exp(-gamma * np.maximum(d2, 0.0)).sum())

    @staticmethod
    def ks_extremes(real, synth, nf):
        from oracle import validation_oracle as V
        ex = [V.ks_count_extremes(real[:, f].numpy(), synth[:, f].numpy()) for f in range(nf)]
        return np.array([e[0] for e in ex], dtype=np.int64), np.array([e[1] for e in ex], dtype=np.int64)

    @staticmethod
    def col_moments(t, cols):
        x = t.double().numpy()[:, cols]
        return x.sum(0), (x * x).sum(0)

    @staticmethod
    def rowz_sq(t, cols, mu, isd):
        z = (t.double().numpy()[:, cols] - mu) * isd
        return float((z.sum(1) ** 2).sum())

    @staticmethod
    def pearson_sums(ta, ca, tb, cb):
        x, y = ta.double().numpy()[:, ca], tb.double().numpy()[:, cb]
        return np.array([x.sum(), y.sum(), (x * x).sum(), (y * y).sum(), (x * y).sum()])


def _val_data():
    rs = np.random.RandomState(4)
    real = torch.from_numpy(rs.randn(60, 24).astype(np.float32))
    synth = torch.from_numpy((rs.randn(101, 24) * 1.1 + 0.1).astype(np.float32))     # 101 rows: ragged shards
    synth[:, 3] = torch.round(synth[:, 3])                                             # ties
    return real, synth


def _val_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import validation_oracle as V
        from osteosarcoma_diffusionmodel_amd.parallel import ShardComm
        from osteosarcoma_diffusionmodel_amd.validation import (sharded_ks_extremes, sharded_mean_offdiag, sharded_mmd,
                                                                 sharded_pearson)
        real, synth = _val_data()
        off, cnt = shard_rows(synth.shape[0], rank, world)
        local = synth[off:off + cnt].contiguous()
        comm, k = ShardComm(True), NumpyKernels()
        assert comm.on and comm.world == world
        ok = True
        gathered = comm.gather_rows(local)
        ok &= torch.equal(gathered, synth)
        ok &= all(torch.equal(s, synth[slice(*(lambda o, c: (o, o + c))(*shard_rows(synth.shape[0], r, world)))]) for r, s in enumerate(comm.shards(local)))
        ok &= abs(sharded_mmd(comm, k, real, local, 1.0 / 24) - V.mmd_rbf(real.numpy(), synth.numpy())) < 1e-9
        dmax, dmin, n2 = sharded_ks_extremes(comm, k, real, local, 10)
        ref = [V.ks_count_extremes(real[:, f].numpy(), synth[:, f].numpy()) for f in range(10)]
        ok &= n2 == 101 and dmax.tolist() == [r[0] for r in ref] and dmin.tolist() == [r[1] for r in ref]
        cols = [1, 4, 5, 9, 20]
        ok &= abs(sharded_mean_offdiag(comm, k, local, cols) - V.mean_offdiag_correlation(synth.numpy(), cols)) < 1e-9
        ok &= abs(sharded_pearson(comm, k, local, 2, local, 7) - V.pearson(synth.numpy()[:, 2], synth.numpy()[:, 7])) < 1e-9
        ok &= comm.bcast_object(("x", rank)) == ("x", 0)
        ok &= comm.sum(np.array([rank + 1, 10], dtype=np.int64)).tolist() == [world * (world + 1) // 2, 10 * world]
        q.put((rank, bool(ok)))
    except Exception as e:                           # report instead of leaving the parent to time out
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])       # 3: every pair met once with weight 2; 4: the half-way round pairs both ends, weight 1
def test_sharded_validation_combination_gloo(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_val_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


def test_shardcomm_inactive_is_identity():
    from osteosarcoma_diffusionmodel_amd.parallel import ShardComm
    c = ShardComm(True)                              # no process group: behaves as world size 1
    t = torch.arange(6.0).view(3, 2)
    assert not c.on and c.world == 1
    assert c.gather_rows(t) is t and list(c.shards(t))[0] is t
    assert c.sum(5).tolist() == [5] and c.bcast_object("a") == "a"


def _prep_worker(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from osteosarcoma_diffusionmodel_amd.train import prepare_data
        conf = {"data": {"processed_dir": root}, "model": {},
                "training": {"batch_size": 8, "val_split": 0.2, "random_seed": 42}}
        train_loader, val_loader, conf = prepare_data(conf)
        seen = []
        for batch in train_loader:
            seen.append(batch["data"][:, 6:16].clone())         # the expression block: real-valued, identifies a row
        rows = torch.cat(seen)
        everyone = [None] * world
        dist.all_gather_object(everyone, rows.numpy().tolist())
        q.put((rank, len(train_loader), everyone if rank == 0 else None, conf["model"]["n_conditions"]))
    finally:
        dist.destroy_process_group()


def test_prepare_data_shards_the_train_split_gloo_world2(tmp_path):
    """utils/train.py:342-444 under data parallel: every rank loads the same files and makes the same seeded train /
    validation split; the DistributedSampler then hands each rank its own half of the train rows per epoch (batch_size is
    per rank), so the ranks' batches are disjoint."""
    import pandas as pd
    rng = np.random.RandomState(0)
    n, root = 100, tmp_path / "processed"
    root.mkdir()
    ids = [f"TARGET-40-{i:04d}" for i in range(n)]
    pd.DataFrame(rng.randint(0, 2, (n, 6)), index=ids, columns=[f"G{i}" for i in range(6)]).to_csv(root / "mutation_matrix_aligned.csv")
    pd.DataFrame(rng.randn(n, 10), index=ids, columns=[f"E{i}" for i in range(10)]).to_csv(root / "expression_matrix_aligned.csv")
    pd.DataFrame(rng.randn(n, 4), index=ids, columns=[f"P{i}" for i in range(4)]).to_csv(root / "pathway_scores.csv")
    pd.DataFrame({"submitter_id": ids, "survival_days": rng.randint(100, 2000, n), "event_occurred": rng.randint(0, 2, n),
                  "age_years": rng.uniform(10, 18, n)}).to_csv(root / "clinical_aligned.csv", index=False)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_prep_worker, args=(r, 2, port, str(root), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(30)
    assert res[0][1] == res[1][1] == 5                     # 80 train rows -> 40 per rank -> 5 batches of 8
    assert res[0][3] == 3                                  # survival_days_norm, event_occurred, age_years
    everyone = res[0][2]
    assert len(everyone[0]) == len(everyone[1]) == 40
    r0 = {tuple(np.round(v, 5)) for v in everyone[0]}
    r1 = {tuple(np.round(v, 5)) for v in everyone[1]}
    assert len(r0) == len(r1) == 40 and not (r0 & r1)      # disjoint halves of the 80-row train split

