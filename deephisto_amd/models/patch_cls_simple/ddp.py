"""Bucketed, overlapped gradient all-reduce for data-parallel training.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  The native
engines lay their gradient arena out so that the backward pass completes it bucket by bucket and call back as soon
as the kernels finishing a bucket are enqueued (`dh_resnet18_set_buckets` / `dh_train2_set_buckets`).  `BucketReducer`
turns each callback into an asynchronous all-reduce of that arena slice on a side stream -- ordered behind the
backward kernels by an event -- so all but the last bucket's exchange hides under the rest of the backward pass;
`finish()` makes the compute stream wait for every bucket and turns the sums into means.  xGMI is point-to-point
(7 links per GPU), so the ring all-reduce is per-link bound: ~25 MB buckets keep each collective bandwidth-bound
without delaying the first launch.

Wire format: float32 by default (bucketed == one collective, bit for bit).  `wire="bf16"` (or DH_DDP_WIRE=bf16) packs every bucket to bf16
on the communication stream (`dh_grad_pack_bf16`, round to nearest even), all-reduces the bf16 copy -- half the bytes per xGMI link -- and
`finish()` unpacks the sums times 1 / world into the float32 arena (`dh_grad_unpack_bf16`): every rank holds the same bf16 sums, so the
replicas stay identical; the averaged gradient carries a relative rounding error of <= 2^-8 per term, as in any bf16 gradient exchange.

Overlap switch: `overlap=False` (or DH_DDP_OVERLAP=0) keeps the SAME buckets and the same arithmetic but starts them one after the other on
the compute stream AFTER the backward pass (`finish()`), so nothing runs beside the persistent one-workgroup-per-CU convolution kernels: the
fall-back -- and the A/B -- for a node where RCCL's kernels do not get CUs while those kernels hold the chip (bench.py reports `train_ddp` both
ways).  Gradients are bit-equal in both modes (`tests/test_ddp.py`).

The reference has no multi-GPU path (single process, train.py:59-301); semantics follow torch DDP: per-rank batch
statistics, gradients averaged, replicas stay identical.
"""
from __future__ import annotations

import os

import torch

DEFAULT_BUCKET_BYTES = 25 * 1024 * 1024


def default_wire() -> str:
    w = os.environ.get("DH_DDP_WIRE", "f32").lower()
    if w not in ("f32", "bf16"):
        raise ValueError(f"DH_DDP_WIRE={w!r}: expected f32 or bf16")
    return w


def default_overlap() -> bool:
    v = os.environ.get("DH_DDP_OVERLAP", "1")
    if v not in ("0", "1"):
        raise ValueError(f"DH_DDP_OVERLAP={v!r}: expected 0 or 1")
    return v == "1"


def allreduce_mean_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place mean of `flat` over the ranks of `group`: ONE collective (the un-bucketed exchange)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
    return flat


def broadcast_state_(model, group=None, src: int = 0, buffers_only: bool = False) -> None:
    """Rank `src`'s parameters and buffers (or the buffers alone: BN running statistics and batch counters) to every rank of
    `group`, the broadcast torch DDP does in its constructor (and, for buffers, before every forward).  The models here are
    initialised independently per rank (unseeded `kaiming_normal_`, torchvision's scheme): without this step the replicas
    would average gradients taken at different points and never become identical.  One flat float32 collective for the
    floating-point tensors and one int64 collective for the counters; the result goes back through `load_state_dict`, so
    the native engines see it like any loaded checkpoint."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        return
    sd = model.state_dict()
    if buffers_only:
        keep = {n for n, _ in model.named_buffers()}
        sd = {k: v for k, v in sd.items() if k in keep}
    if not sd:
        return
    on_gpu = dist.get_backend(group) == "nccl"
    fl = [k for k, v in sd.items() if v.is_floating_point()]
    it = [k for k, v in sd.items() if not v.is_floating_point()]
    for keys, dtype in ((fl, torch.float32), (it, torch.int64)):
        if not keys:
            continue
        flat = torch.cat([sd[k].detach().to(dtype).reshape(-1) for k in keys])
        home = flat.device
        if on_gpu and not flat.is_cuda:
            flat = flat.cuda()
        dist.broadcast(flat, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
        flat, off = flat.to(home), 0
        for k in keys:
            n = sd[k].numel()
            sd[k] = flat[off:off + n].reshape(sd[k].shape).to(sd[k].dtype)
            off += n
    model.load_state_dict(sd, strict=not buffers_only)


class BucketReducer:
    """Asynchronous per-bucket all-reduce of slices of one flat gradient tensor.

    `on_bucket(bucket, offset, count)` may be called from a native callback while the producer is still enqueueing
    work; on CUDA tensors the collective is ordered behind everything enqueued so far on the current stream."""

    def __init__(self, flat: torch.Tensor, group=None, wire: str | None = None, overlap: bool | None = None):
        import torch.distributed as dist

        self.flat, self.group = flat, group
        self.overlap = default_overlap() if overlap is None else bool(overlap)
        self.deferred = []   # overlap off: (offset, count) of the buckets reported so far; exchanged by finish()
        self.world = dist.get_world_size(group)
        self.works = []
        self.log = []   # (bucket, offset, count) in launch order
        self.cuda = flat.is_cuda
        self.wire = wire or default_wire()
        if self.wire not in ("f32", "bf16"):
            raise ValueError(f"wire={self.wire!r}: expected f32 or bf16")
        self.staging = _wire_buffer(flat) if self.wire == "bf16" else None
        if self.cuda:
            self.main = torch.cuda.current_stream(flat.device)
            self.comm = _comm_stream(flat.device)

    def _pack(self, offset: int, count: int) -> torch.Tensor:
        """bf16 copy of arena[offset : offset + count] in the staging buffer (on the current stream)."""
        dst = self.staging[offset:offset + count]
        if self.cuda:
            import ctypes as C

            from ..._lib import check, lib
            st = C.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream)
            check(lib().dh_grad_pack_bf16(self.flat.data_ptr() + 4 * offset, dst.data_ptr(), count, st), "dh_grad_pack_bf16")
        else:   # host tensors only occur in the gloo tests of this helper
            dst.copy_(self.flat[offset:offset + count])
        return dst

    def on_bucket(self, bucket: int, offset: int, count: int) -> None:
        import torch.distributed as dist

        self.log.append((int(bucket), int(offset), int(count)))
        if not self.overlap:          # exchanged after the backward pass, in this order (finish)
            self.deferred.append((int(offset), int(count)))
            return
        bf16 = self.wire == "bf16"
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(self.main)             # everything the producer has enqueued so far
            self.comm.wait_event(ev)
            with torch.cuda.stream(self.comm):
                piece = self._pack(offset, count) if bf16 else self.flat[offset:offset + count]
                self.works.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            piece = self._pack(offset, count) if bf16 else self.flat[offset:offset + count]
            self.works.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> torch.Tensor:
        """Wait for every bucket (the compute stream waits on the device; nothing blocks the host with nccl) and divide."""
        import torch.distributed as dist

        for offset, count in self.deferred:   # overlap off: the same buckets, in completion order, on the compute stream
            piece = self._pack(offset, count) if self.wire == "bf16" else self.flat[offset:offset + count]
            dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group)
        self.deferred = []
        for w in self.works:
            w.wait()
        if self.cuda and self.overlap:
            self.main.wait_stream(self.comm)
        covered = sum(c for _, _, c in self.log)
        if covered != self.flat.numel():
            raise RuntimeError(f"gradient buckets covered {covered} of {self.flat.numel()} elements")
        if self.wire == "bf16":
            if self.cuda:
                import ctypes as C

                from ..._lib import check, lib
                st = C.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream)
                check(lib().dh_grad_unpack_bf16(self.staging.data_ptr(), self.flat.data_ptr(), self.flat.numel(), 1.0 / self.world, st),
                      "dh_grad_unpack_bf16")
            else:
                self.flat.copy_(self.staging.float() * (1.0 / self.world))
        else:
            self.flat.div_(self.world)
        self.works = []
        return self.flat


_COMM_STREAMS = {}
_WIRE_BUFFERS = {}


def _wire_buffer(flat: torch.Tensor) -> torch.Tensor:
    """bf16 staging copy of a gradient arena, kept per (device, size): the exchange of every step reuses it."""
    key = (str(flat.device), flat.numel())
    buf = _WIRE_BUFFERS.get(key)
    if buf is None:
        buf = _WIRE_BUFFERS[key] = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
    return buf


def _comm_stream(device) -> "torch.cuda.Stream":
    key = torch.device(device).index
    if key not in _COMM_STREAMS:
        _COMM_STREAMS[key] = torch.cuda.Stream(device)
    return _COMM_STREAMS[key]
