"""One process per GPU: process-group setup for the two multi-GPU entry points.

`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P -m
models.patch_cls_simple.train` (BASELINE configs[4]: data-parallel training, RCCL all-reduce of gradient buckets) and
`... -m examples.predict_full_patched` (configs[3]: tile ranges sharded over the ranks, ONE RCCL all-gather of logits).
The reference is single-process (`models/patch_cls_simple/train.py:304-315`, `examples/predict_full_patched.py:128-177`);
these helpers are what turns its two `__main__` blocks into per-rank programs.

Order matters on ROCm: the rank binds its GPU (`torch.cuda.set_device(LOCAL_RANK)`) and joins the group
(`init_process_group("nccl", device_id=...)`: backend "nccl" IS RCCL) BEFORE any other GPU call, so no rank ever creates a
context on cuda:0 by accident.  `DH_DIST_BACKEND=gloo` selects the CPU backend (tests, rehearsals on a one-GPU box together
with `DH_SHARE_GPU=1`, which maps every rank to cuda:0).

Timeouts: the group is created with an explicit collective timeout (`DH_DIST_TIMEOUT_S`, default 600 s, instead of the backend's
own default of 10-30 minutes).  A rank whose peer is stuck then fails inside the collective -- RCCL's watchdog aborts the process,
gloo raises -- and leaves through `finalize(owned, ok=False)`: non-zero exit, no barrier, no re-exec; the launcher tears the job down.
"""
from __future__ import annotations

import os


def env_world() -> tuple[int, int, int]:
    """(rank, world, local_rank) from the launcher's environment (1 process when absent)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")))


def dist_timeout():
    """Collective timeout of the process groups created here (DH_DIST_TIMEOUT_S seconds, default 600)."""
    from datetime import timedelta

    v = os.environ.get("DH_DIST_TIMEOUT_S", "600")
    try:
        sec = float(v)
    except ValueError:
        raise ValueError(f"DH_DIST_TIMEOUT_S={v!r}: expected seconds") from None
    if not sec > 0:
        raise ValueError(f"DH_DIST_TIMEOUT_S={v!r}: expected a positive number of seconds")
    return timedelta(seconds=sec)


def init_from_env():
    """Join the process group torch.distributed.run describes.  Returns (rank, world, device_index or None, owned):
    `owned` says this call created the group (the caller then ends with `finalize(owned)`).  Single process: no-op."""
    import torch
    import torch.distributed as dist

    rank, world, local = env_world()
    if world <= 1:
        return 0, 1, (torch.cuda.current_device() if torch.cuda.is_available() else None), False
    if dist.is_initialized():   # an embedding program (bench.py, a test) already did it
        return dist.get_rank(), dist.get_world_size(), (torch.cuda.current_device() if torch.cuda.is_available() else None), False
    backend = os.environ.get("DH_DIST_BACKEND", "nccl")
    dev_index = None
    if backend == "nccl" or os.environ.get("DH_SHARE_GPU") == "1":
        dev_index = 0 if os.environ.get("DH_SHARE_GPU") == "1" else local
        torch.cuda.set_device(dev_index)           # before anything else touches a GPU
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=dist_timeout())   # RCCL over xGMI
    else:
        dist.init_process_group(backend, timeout=dist_timeout())
    return dist.get_rank(), dist.get_world_size(), dev_index, True


def finalize(owned: bool, ok: bool = True) -> None:
    """End of a per-rank program.  Success: barrier, then destroy the group.  `ok=False` (the rank is leaving through an
    exception): NO collective -- its peers sit in some other collective (a bucket all-reduce, the all-gather), a barrier here
    would mismatch with it and block until the backend's timeout, hiding the exception; the process exits non-zero instead and
    `torch.distributed.run` kills the other ranks."""
    import torch.distributed as dist

    if not (owned and dist.is_initialized()):
        return
    if ok:
        dist.barrier()
        dist.destroy_process_group()
    else:
        try:   # release the communicator without synchronising with anyone; never mask the original exception
            abort = getattr(dist.distributed_c10d, "_abort_process_group", None)
            if abort is not None and dist.get_backend() == "nccl":
                abort()
        except Exception:
            pass
