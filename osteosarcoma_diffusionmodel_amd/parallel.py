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


class RcclGradComm:
    """The library's own RCCL communicator (osd_comm_*, include/osdiff.h): the gradient all-reduce runs from C on
    the communicator's stream, bucket by bucket behind the events of osd_train_loss_fwd_bwd, with no Python in the
    loop.  The 128-byte rendezvous id is drawn by rank 0 and shipped through torch.distributed (any backend)."""

    def __init__(self, device: torch.device):
        import ctypes as C
        from . import _lib as L
        self._L, self._C = L, C
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        box = [None]
        if self.rank == 0:
            buf = (C.c_char * L.OSD_COMM_ID_BYTES)()
            L.check(L.lib().osd_comm_unique_id(buf))
            box[0] = bytes(buf.raw)
        dist.broadcast_object_list(box, src=0)
        uid = (C.c_char * L.OSD_COMM_ID_BYTES).from_buffer_copy(box[0])
        self.handle = C.c_void_p()
        dev = device.index if device.index is not None else torch.cuda.current_device()
        L.check(L.lib().osd_comm_create(uid, self.rank, self.world, dev, C.byref(self.handle)))
        self._starts = self._ends = None

    def set_buckets(self, slices: Sequence[Tuple[int, int]]):
        C = self._C
        n = len(slices)
        self._starts = (C.c_int64 * n)(*[s for s, _ in slices])
        self._ends = (C.c_int64 * n)(*[e for _, e in slices])
        self._n = n

    def allreduce(self, engine_handle, flat_grad: torch.Tensor, events=None):
        """begin + end: the handle's stream continues once every bucket is reduced."""
        L, C = self._L, self._C
        ev = None
        if events is not None:
            ev = (C.c_void_p * self._n)(*[e.cuda_event for e in events])
        L.check(L.lib().osd_allreduce_grads_begin(engine_handle, self.handle, L.ptr(flat_grad), self._starts, self._ends, ev, self._n))
        L.check(L.lib().osd_allreduce_grads_end(engine_handle, self.handle))

    def close(self):
        if self.handle:
            self._L.lib().osd_comm_destroy(self.handle)
            self.handle = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardComm:
    """The exchanges of row-sharded validation (third row): every rank holds a shard of the
    synthetic patients; metric accumulators are summed, the few columns a sort needs are gathered, and the
    all-pairs Gram block walks the shards by broadcast.  Inactive (world size 1 semantics) unless ``enabled`` and
    ``torch.distributed`` is initialised.  Works on RCCL ("nccl", device tensors) and on gloo (staged through the
    host), which is what the CPU tests drive."""

    def __init__(self, enabled: bool = True):
        self.on = bool(enabled) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if self.on else 1
        self.rank = dist.get_rank() if self.on else 0
        self._nccl = self.on and dist.get_backend() == "nccl"

    def _stage(self, t: torch.Tensor) -> torch.Tensor:
        if self._nccl:
            return t if t.is_cuda else t.cuda()
        return t.cpu()

    def sum(self, values):
        """Element-wise SUM over ranks of a float64 / int64 numpy array (or python scalar); returns numpy."""
        import numpy as np
        arr = np.atleast_1d(np.asarray(values))
        if not self.on:
            return arr.copy()
        t = self._stage(torch.from_numpy(np.ascontiguousarray(arr)))
        dist.all_reduce(t)
        return t.cpu().numpy()

    def gather_rows(self, t: torch.Tensor) -> torch.Tensor:
        """Concatenation over ranks (rank order) of [n_r, cols] tensors with ragged n_r, on t's device."""
        if not self.on:
            return t
        counts = self.sum(_one_hot(self.rank, self.world) * t.shape[0]).astype("int64")
        nmax = int(counts.max())
        pad = torch.zeros((nmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        staged = self._stage(pad.contiguous())
        parts = [torch.empty_like(staged) for _ in range(self.world)]
        dist.all_gather(parts, staged)
        return torch.cat([p[: int(c)] for p, c in zip(parts, counts)], dim=0).to(t.device)

    def shards(self, t: torch.Tensor):
        """Yield every rank's [n_r, cols] tensor in rank order on t's device (one broadcast per rank, so the
        working set is one shard, not the whole population)."""
        if not self.on:
            yield t
            return
        counts = self.sum(_one_hot(self.rank, self.world) * t.shape[0]).astype("int64")
        for src in range(self.world):
            if src == self.rank:
                buf = self._stage(t.contiguous())
            else:
                buf = self._stage(torch.empty((int(counts[src]),) + tuple(t.shape[1:]), dtype=t.dtype,
                                              device=t.device if self._nccl else "cpu"))
            dist.broadcast(buf, src=src)
            yield t if src == self.rank else buf.to(t.device)

    def ring_partners(self, t: torch.Tensor):
        """Yield (shard, weight) so that every UNORDERED pair of different ranks meets exactly once over the whole group: in round
        d = 1 .. world // 2 rank r sends its [n_r, cols] tensor to rank r - d and receives rank (r + d)'s (point to point: half the
        shard traffic of ``shards``, which broadcasts every shard to everybody).  Weight 2 = the pair (r, r + d) is this rank's alone;
        the last round of an even world pairs r with r + world / 2 from both ends, each end taking weight 1."""
        if not self.on:
            return
        counts = self.sum(_one_hot(self.rank, self.world) * t.shape[0]).astype("int64")
        mine = self._stage(t.contiguous())
        for d in range(1, self.world // 2 + 1):
            src, dst = (self.rank + d) % self.world, (self.rank - d) % self.world
            buf = torch.empty((int(counts[src]),) + tuple(t.shape[1:]), dtype=t.dtype, device=mine.device)
            ops = [dist.P2POp(dist.isend, mine, dst), dist.P2POp(dist.irecv, buf, src)]
            if self.rank > dst:                      # lower rank of a pair posts its receive first (deadlock-free order on gloo)
                ops.reverse()
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            yield buf.to(t.device), (1.0 if 2 * d == self.world else 2.0)

    def bcast_object(self, obj, src: int = 0):
        if not self.on:
            return obj
        box = [obj]
        dist.broadcast_object_list(box, src=src)
        return box[0]


def _one_hot(i: int, n: int):
    import numpy as np
    v = np.zeros(n, dtype=np.int64)
    v[i] = 1
    return v
