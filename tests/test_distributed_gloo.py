"""N>1 path of predict_full_patched on CPU: two gloo ranks shard the reference-ordered
tile list, exchange per-tile logits with the same all-gather code the GPU path uses,
and must rebuild exactly the single-process result (CPU only, world_size 2)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tiling


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _toy_logits(origins):
    o = origins.astype(np.float64)
    return np.stack([np.sin(o[:, 0] * 0.01 + k) + np.cos(o[:, 1] * 0.013 * (k + 1)) for k in range(5)], 1).astype(np.float32)


def _worker(rank, world, port, h, w, P, S, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deephisto_amd.examples.predict_full_patched import exchange_logits, shard_range
    o = tiling.tile_origins(h, w, P, S)
    n = len(o)
    lo, hi = shard_range(n, world, rank)
    per_rank = -(-n // world)
    local = torch.zeros((per_rank, 5))
    local[:hi - lo] = torch.from_numpy(_toy_logits(o[lo:hi]))
    full = exchange_logits(local, n)
    q.put((rank, lo, hi, full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("h,w,P,S,B", [(1000, 1300, 256, 256, 16), (777, 1033, 100, 37, 7)])
def test_sharded_exchange_matches_single_process(h, w, P, S, B):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, h, w, P, S, B, q)) for r in range(world)]
    [p.start() for p in procs]
    got = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    o = tiling.tile_origins(h, w, P, S)
    want = _toy_logits(o)
    ranges = sorted((lo, hi) for _, lo, hi, _ in got)
    assert ranges[0][0] == 0 and ranges[-1][1] == len(o) and ranges[0][1] == ranges[1][0]
    for _, _, _, full in got:
        assert full.shape == want.shape
        assert np.array_equal(full, want)           # bit-identical on every rank
    # downstream: padded list + accumulation equals the single-process oracle result
    padded = tiling.batched_origins(h, w, P, S, B).reshape(-1, 2)
    pad = len(padded) - len(o)
    logits = np.concatenate([want, np.repeat(want[-1:], pad, 0)])
    canvas = tiling.accumulate_logits(h, w, 5, 16, P, padded, logits)
    canvas2 = tiling.accumulate_logits(h, w, 5, 16, P, padded, np.concatenate([got[0][3], np.repeat(got[0][3][-1:], pad, 0)]))
    assert np.array_equal(canvas, canvas2)


def test_shard_range_partition():
    from deephisto_amd.examples.predict_full_patched import shard_range
    for n in (0, 1, 7, 38416, 198916):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(38416, 8, 3) == (3 * 4802, 4 * 4802)
