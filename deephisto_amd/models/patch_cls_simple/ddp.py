"""Bucketed, overlapped gradient all-reduce for data-parallel training.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  The native
engines lay their gradient arena out so that the backward pass completes it bucket by bucket and call back as soon
as the kernels finishing a bucket are enqueued (`dh_resnet18_set_buckets` / `dh_train2_set_buckets`).  `BucketReducer`
turns each callback into an asynchronous all-reduce of that arena slice on a side stream -- ordered behind the
backward kernels by an event -- so all but the last bucket's exchange hides under the rest of the backward pass;
`finish()` makes the compute stream wait for every bucket and turns the sums into means.  xGMI is point-to-point
(7 links per GPU), so the ring all-reduce is per-link bound: ~25 MB buckets keep each collective bandwidth-bound
without delaying the first launch.

The reference has no multi-GPU path (single process, train.py:59-301); semantics follow torch DDP: per-rank batch
statistics, gradients averaged, replicas stay identical.
"""
from __future__ import annotations

import torch

DEFAULT_BUCKET_BYTES = 25 * 1024 * 1024


def allreduce_mean_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place mean of `flat` over the ranks of `group`: ONE collective (the un-bucketed exchange)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
    return flat


class BucketReducer:
    """Asynchronous per-bucket all-reduce of slices of one flat gradient tensor.

    `on_bucket(bucket, offset, count)` may be called from a native callback while the producer is still enqueueing
    work; on CUDA tensors the collective is ordered behind everything enqueued so far on the current stream."""

    def __init__(self, flat: torch.Tensor, group=None):
        import torch.distributed as dist

        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group)
        self.works = []
        self.log = []   # (bucket, offset, count) in launch order
        self.cuda = flat.is_cuda
        if self.cuda:
            self.main = torch.cuda.current_stream(flat.device)
            self.comm = _comm_stream(flat.device)

    def on_bucket(self, bucket: int, offset: int, count: int) -> None:
        import torch.distributed as dist

        piece = self.flat[offset:offset + count]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(self.main)             # everything the producer has enqueued so far
            self.comm.wait_event(ev)
            with torch.cuda.stream(self.comm):
                self.works.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.works.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.log.append((int(bucket), int(offset), int(count)))

    def finish(self) -> torch.Tensor:
        """Wait for every bucket (the compute stream waits on the device; nothing blocks the host with nccl) and divide."""
        for w in self.works:
            w.wait()
        if self.cuda:
            self.main.wait_stream(self.comm)
        covered = sum(c for _, _, c in self.log)
        if covered != self.flat.numel():
            raise RuntimeError(f"gradient buckets covered {covered} of {self.flat.numel()} elements")
        self.flat.div_(self.world)
        self.works = []
        return self.flat


_COMM_STREAMS = {}


def _comm_stream(device) -> "torch.cuda.Stream":
    key = torch.device(device).index
    if key not in _COMM_STREAMS:
        _COMM_STREAMS[key] = torch.cuda.Stream(device)
    return _COMM_STREAMS[key]
