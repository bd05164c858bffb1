"""Data-parallel training (row e): gradient averaging over ranks.
CPU part (gloo, world_size 2): the collective helper.  GPU part (two processes on cuda:0,
gloo transport): fused HIP train steps stay replica-identical and match two oracle replicas
whose gradients are averaged by hand."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cpu_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deephisto_amd.models.patch_cls_simple.model import allreduce_mean_
    t = torch.arange(10, dtype=torch.float32) * (rank + 1)
    allreduce_mean_(t)
    q.put((rank, t.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_mean_gloo_cpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_cpu_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    got = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    want = np.arange(10, dtype=np.float32) * 1.5
    for _, a in got:
        np.testing.assert_allclose(a, want, rtol=0, atol=0)


def _gpu_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deephisto_amd.models.patch_cls_simple.model import get_model
    from oracle import resnet18 as oracle_net
    dev = torch.device("cuda:0")
    ref = oracle_net.seeded_model(21, 5, perturb_bn=True)
    m = get_model(5, "f32")
    m.load_state_dict(ref.state_dict())
    m.to(dev).train()
    g = torch.Generator().manual_seed(100 + rank)   # every rank its own data stream
    losses = []
    for _ in range(2):
        x = torch.rand(4, 3, 64, 64, generator=g)
        y = torch.randint(0, 5, (4,), generator=g)
        loss, _ = m.train_step(x.to(dev), y.to(dev), lr=1e-4)
        losses.append(float(loss))
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    q.put((rank, losses, sd["conv1.weight"].numpy(), sd["fc.weight"].numpy(), sd["layer3.0.downsample.0.weight"].numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_ddp_two_ranks_match_averaged_oracle(built_lib):
    import torch.nn.functional as F
    from oracle import resnet18 as oracle_net
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    [p.join(timeout=120) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    # replicas identical after the averaged updates
    for a, b in zip(got[0][2:], got[1][2:]):
        assert np.array_equal(a, b)
    # oracle: two replicas, gradients averaged by hand, same Adam
    reps = [oracle_net.seeded_model(21, 5, perturb_bn=True).train() for _ in range(world)]
    opts = [torch.optim.Adam(r.parameters(), lr=1e-4) for r in reps]
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    want_losses = [[], []]
    for _ in range(2):
        for r in range(world):
            x = torch.rand(4, 3, 64, 64, generator=gens[r])
            y = torch.randint(0, 5, (4,), generator=gens[r])
            opts[r].zero_grad()
            loss = F.cross_entropy(reps[r](x), y)
            loss.backward()
            want_losses[r].append(float(loss))
        for ps_ in zip(*[list(r.parameters()) for r in reps]):
            avg = sum(p.grad for p in ps_) / world
            for p in ps_:
                p.grad = avg.clone()
        [o.step() for o in opts]
    for r in range(world):
        assert abs(got[r][1][0] - want_losses[r][0]) <= 1e-4
        assert abs(got[r][1][1] - want_losses[r][1]) <= 1e-3
    sd = reps[0].state_dict()
    for a, k in zip(got[0][2:], ["conv1.weight", "fc.weight", "layer3.0.downsample.0.weight"]):
        d = np.abs(a - sd[k].numpy())
        assert d.max() <= 4e-4 and d.mean() <= 2e-5, (k, d.max(), d.mean())
