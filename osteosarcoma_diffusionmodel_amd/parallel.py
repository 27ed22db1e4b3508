"""Data-parallel host logic: how patients and gradient buckets are split over ranks.

One process per GPU.  Reverse sampling shards patients with NO collective; training
all-reduces one flat gradient buffer, bucket by bucket, over RCCL (backend "nccl" on
ROCm) -- or gloo in the CPU tests, which exercise exactly these functions.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_rows(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced shard of n rows for `rank`: (global offset, count).  The offset is
    what a rank passes as ``row_offset`` so that Philox draws are independent of the GPU count."""
    base, rem = divmod(n, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def bucket_slices(offsets: Sequence[int], buckets: Sequence[Tuple[int, int]]) -> List[Tuple[int, int]]:
    """Flat-buffer [start, end) of each gradient bucket (parameters first..last inclusive)."""
    return [(int(offsets[f]), int(offsets[l + 1])) for f, l in buckets]


def allreduce_buckets(flat_grad: torch.Tensor, slices: Sequence[Tuple[int, int]], events=None, comm_stream=None):
    """SUM all-reduce of each bucket slice, in backward order.  On GPU, bucket b is enqueued on
    ``comm_stream`` behind ``events[b]`` (recorded by osd_train_loss_fwd_bwd when that bucket's
    gradients are final), which overlaps the collective with the rest of backward.  Gradients
    are pre-scaled by 1/world in the loss, so the SUM is the data-parallel mean."""
    if comm_stream is None:
        for s, e in slices:
            dist.all_reduce(flat_grad[s:e])
        return
    main = torch.cuda.current_stream()
    with torch.cuda.stream(comm_stream):
        for (s, e), ev in zip(slices, events):
            comm_stream.wait_event(ev)
            dist.all_reduce(flat_grad[s:e])
    main.wait_stream(comm_stream)
