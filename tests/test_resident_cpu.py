"""CPU: the device-resident epoch path's batch order (ResidentSplit.epoch_indices) is the DataLoader's own, draw for draw --
shuffled + drop_last training split, sequential validation split, DistributedSampler shard -- and every row of the split is
visited exactly once per epoch (utils/train.py:204-250, 413-437)."""
import numpy as np
import torch
from torch.utils.data import DataLoader

from osteosarcoma_diffusionmodel_amd.train import OsteosarcomaDataset, ResidentSplit


def _dataset(n=103, d=6):
    ds = object.__new__(OsteosarcomaDataset)
    ds.data = torch.arange(n, dtype=torch.float32)[:, None].repeat(1, d)
    ds.conditions = torch.arange(n, dtype=torch.float32)[:, None].repeat(1, 3) * 10
    ds.survival_days = torch.arange(n, dtype=torch.float32)
    return ds


def _visited(loader):
    return [b["data"][:, 0].to(torch.int64) for b in loader]


def test_order_matches_the_dataloader_shuffled_and_sequential():
    base = _dataset()
    train_ds, val_ds = torch.utils.data.random_split(base, [83, 20], generator=torch.Generator().manual_seed(42))
    train = DataLoader(train_ds, batch_size=16, shuffle=True, num_workers=0, drop_last=True)
    val = DataLoader(val_ds, batch_size=16, shuffle=False, num_workers=0)
    cache = {}
    rt = ResidentSplit.build(train, "cpu", cache, budget_bytes=1 << 30)
    rv = ResidentSplit.build(val, "cpu", cache, budget_bytes=1 << 30)
    assert rt is not None and rv is not None and len(cache) == 1          # both splits share one upload of the base dataset
    for epoch in range(3):
        torch.manual_seed(100 + epoch)
        want = _visited(train)
        state = torch.get_rng_state()
        torch.manual_seed(100 + epoch)
        got = rt.epoch_indices()
        assert torch.equal(torch.get_rng_state(), state)                  # the same number of draws from the default generator
        assert len(got) == len(want) == 5
        for a, b in zip(got, want):
            assert torch.equal(rt.base[0][a][:, 0].to(torch.int64), b)
        rows = torch.cat(got)
        assert rows.unique().numel() == rows.numel() == 80                # drop_last: 5 full batches, each row at most once
        assert set(rows.tolist()) <= set(train_ds.indices)
    got_v = rv.epoch_indices()
    assert [g.tolist() for g in got_v] == [b.tolist() for b in _visited(val)]
    assert sorted(torch.cat(got_v).tolist()) == sorted(val_ds.indices)   # every validation row exactly once, last batch short


def test_order_matches_a_loader_with_its_own_generator_over_epochs():
    """DataLoader(shuffle=True, generator=g): RandomSampler draws TWO permutations per epoch from g (the pass and its empty tail),
    so epoch 2 only matches if the resident path makes the second draw as well (round-3 advisor finding)."""
    base = _dataset(90)
    want_loader = DataLoader(base, batch_size=16, shuffle=True, num_workers=0, drop_last=True, generator=torch.Generator().manual_seed(5))
    got_loader = DataLoader(base, batch_size=16, shuffle=True, num_workers=0, drop_last=True, generator=torch.Generator().manual_seed(5))
    rs = ResidentSplit.build(got_loader, "cpu", {}, budget_bytes=1 << 30)
    assert rs is not None
    for epoch in range(4):
        want = _visited(want_loader)
        got = rs.epoch_indices()
        assert len(got) == len(want) == 5
        for a, b in zip(got, want):
            assert torch.equal(rs.base[0][a][:, 0].to(torch.int64), b), f"epoch {epoch}"
        assert torch.equal(want_loader.generator.get_state(), got_loader.generator.get_state())


def test_order_matches_a_distributed_sampler_shard():
    base = _dataset(200)
    train_ds, _ = torch.utils.data.random_split(base, [160, 40], generator=torch.Generator().manual_seed(1))
    for rank in range(2):
        smp = torch.utils.data.distributed.DistributedSampler(train_ds, num_replicas=2, rank=rank, shuffle=True, seed=7, drop_last=True)
        loader = DataLoader(train_ds, batch_size=16, sampler=smp, num_workers=0, drop_last=True)
        rs = ResidentSplit.build(loader, "cpu", {}, budget_bytes=1 << 30)
        for epoch in range(2):
            smp.set_epoch(epoch)
            want = _visited(loader)
            got = rs.epoch_indices()
            assert [g.tolist() for g in got] == [w.tolist() for w in want]


def test_ineligible_loaders_fall_back():
    base = _dataset()
    rows = [{"data": torch.zeros(4), "conditions": torch.zeros(3), "survival": torch.tensor(0.0)} for _ in range(10)]
    assert ResidentSplit.build(DataLoader(rows, batch_size=2), "cpu", {}, budget_bytes=1 << 30) is None            # not an OsteosarcomaDataset
    assert ResidentSplit.build(DataLoader(base, batch_size=2, collate_fn=lambda b: b), "cpu", {}, budget_bytes=1 << 30) is None
    assert ResidentSplit.build(DataLoader(base, batch_size=2), "cpu", {}, budget_bytes=16) is None                 # over the budget
    assert ResidentSplit.build([], "cpu", {}, budget_bytes=1 << 30) is None
