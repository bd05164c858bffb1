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


# ---- bucketed, overlapped all-reduce (models/patch_cls_simple/ddp.py) ---------------------------------------------
def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deephisto_amd.models.patch_cls_simple.ddp import BucketReducer, allreduce_mean_
    g = torch.Generator().manual_seed(7 + rank)
    flat = torch.randn(10_000, generator=g)
    want = allreduce_mean_(flat.clone())
    # buckets arrive in completion order while "backward" is still producing the rest: here the tail of the arena first
    ranges = [(7000, 3000), (2500, 4500), (0, 2500)]
    red = BucketReducer(flat, None)
    for b, (off, cnt) in enumerate(ranges):
        red.on_bucket(b, off, cnt)
    out = red.finish()
    # overlap off (DH_DDP_OVERLAP=0 / overlap=False): same buckets, exchanged one after the other by finish(): same bits
    flat2 = torch.randn(10_000, generator=torch.Generator().manual_seed(7 + rank))
    late = BucketReducer(flat2, None, overlap=False)
    for b, (off, cnt) in enumerate(ranges):
        late.on_bucket(b, off, cnt)
    assert not late.works and len(late.deferred) == 3          # nothing started before finish()
    out_late = late.finish()
    assert torch.equal(out_late, out) and late.log == red.log
    ok_cover = True
    try:
        bad = BucketReducer(torch.zeros(10), None)
        bad.on_bucket(0, 0, 4)
        bad.finish()
        ok_cover = False          # a hole in the bucket cover must be reported
    except RuntimeError:
        pass
    q.put((rank, out.numpy(), want.numpy(), red.log, ok_cover))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_reducer_equals_one_allreduce_gloo_cpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    got = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for _, out, want, log, ok_cover in got:
        np.testing.assert_array_equal(out, want)               # bucketed == one collective, bit for bit (same sums)
        assert [b for b, _, _ in log] == [0, 1, 2] and ok_cover
    np.testing.assert_array_equal(got[0][1], got[1][1])


def _bucket_bf16_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deephisto_amd.models.patch_cls_simple.ddp import BucketReducer, allreduce_mean_
    g = torch.Generator().manual_seed(17 + rank)
    flat = torch.randn(10_000, generator=g)
    want = allreduce_mean_(flat.clone())
    red = BucketReducer(flat, None, wire="bf16")
    for b, (off, cnt) in enumerate([(7000, 3000), (2500, 4500), (0, 2500)]):
        red.on_bucket(b, off, cnt)
    out = red.finish()
    q.put((rank, out.numpy(), want.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_reducer_bf16_wire_gloo_cpu():
    """bf16 wire format of the bucketed exchange (host tensors, gloo): every rank ends with the SAME averaged gradients (the ranks
    reduce identical bf16 sums), within bf16 rounding of the float32 mean."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_bucket_bf16_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    got = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    np.testing.assert_array_equal(got[0][1], got[1][1])
    for _, out, want in got:
        assert np.abs(out - want).max() <= 2.0 ** -7 * np.abs(want).max() + 2.0 ** -8 * 4.0


def _gpu_bf16_wire_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deephisto_amd.models.patch_cls_simple.model import get_model
    from oracle import resnet50 as oracle_net
    dev = torch.device("cuda:0")
    ref = oracle_net.seeded_model(4, 5, perturb_bn=True)
    g = torch.Generator().manual_seed(300 + rank)
    x = torch.rand(4, 3, 64, 64, generator=g).to(dev)
    y = torch.randint(0, 5, (4,), generator=g).to(dev)
    grads = {}
    for wire in ("f32", "bf16"):
        m = get_model(5, arch="resnet50")
        m.load_state_dict(ref.state_dict())
        m.to(dev).train()
        m._engine.ddp_wire = wire
        loss, _ = m.train_step(x, y, lr=1e-4)
        grads[wire] = m.flat_gradients(dev).clone().cpu()
        if wire == "bf16":
            sd = m.state_dict()
            out = (sd["fc.weight"].cpu().numpy(), sd["layer1.0.conv1.weight"].cpu().numpy())
        del m
    rel = float((grads["bf16"] - grads["f32"]).norm() / grads["f32"].norm())
    q.put((rank, rel, grads["bf16"].numpy(), out[0], out[1]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_ddp_bf16_wire_keeps_replicas_identical(built_lib):
    """Two ranks on one GPU (gloo), ResNet-50 bf16 engine, DH_DDP_WIRE=bf16 semantics through `ddp_wire`: the averaged gradients are the
    float32 exchange's within bf16 rounding (relative L2 <= 4e-3) and both replicas hold identical gradients and parameters."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_gpu_bf16_wire_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    [p.join(timeout=120) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for r in range(world):
        assert got[r][1] <= 4e-3, got[r][1]
    assert np.array_equal(got[0][2], got[1][2]) and np.array_equal(got[0][3], got[1][3]) and np.array_equal(got[0][4], got[1][4])


def _gpu_bf16_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deephisto_amd.models.patch_cls_simple.model import get_model
    from oracle import resnet50 as oracle_net
    dev = torch.device("cuda:0")
    ref = oracle_net.seeded_model(4, 5, perturb_bn=True)
    m = get_model(5, arch="resnet50")
    m.load_state_dict(ref.state_dict())
    m.to(dev).train()
    g = torch.Generator().manual_seed(200 + rank)
    x = torch.rand(4, 3, 64, 64, generator=g).to(dev)
    y = torch.randint(0, 5, (4,), generator=g).to(dev)
    # reference: un-bucketed exchange of the same local gradients
    import ctypes as C
    from deephisto_amd._lib import check, lib
    from deephisto_amd.models.patch_cls_simple.ddp import allreduce_mean_
    from deephisto_amd.models.patch_cls_simple.model import ce_loss
    eng = m._engine
    logits = eng.forward(x, True, pull_stats=False)
    _, dl = ce_loss(logits, y, want_grad=True)
    check(lib().dh_train2_backward(eng.handle, dl.data_ptr(), None), "backward")
    want = allreduce_mean_(m.flat_gradients(dev).clone()).cpu()
    # product path: bucketed + overlapped inside train_step (same inputs, same parameters -> same local gradients)
    m2 = get_model(5, arch="resnet50")
    m2.load_state_dict(ref.state_dict())
    m2.to(dev).train()
    loss, _ = m2.train_step(x, y, lr=1e-4, bucket_bytes=25 * 1024 * 1024)
    got = m2.flat_gradients(dev).clone().cpu()     # the arena still holds the averaged gradients after Adam
    log = m2._engine.overlap_log
    sd = m2.state_dict()
    # DH_DDP_OVERLAP=0: the same buckets exchanged after the backward pass -- the fall-back must give the same bits
    os.environ["DH_DDP_OVERLAP"] = "0"
    m3 = get_model(5, arch="resnet50")
    m3.load_state_dict(ref.state_dict())
    m3.to(dev).train()
    loss3, _ = m3.train_step(x, y, lr=1e-4, bucket_bytes=25 * 1024 * 1024)
    late_equal = bool(torch.equal(m3.flat_gradients(dev).cpu(), got)) and float(loss3) == float(loss) and m3._engine.overlap_log == log
    del os.environ["DH_DDP_OVERLAP"]
    q.put((rank, float(loss), bool(torch.equal(got, want)), log, sd["fc.weight"].cpu().numpy(), sd["conv1.weight"].cpu().numpy(), late_equal))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_ddp_resnet50_bf16_bucketed_overlap(built_lib):
    """Two ranks on one GPU (gloo): the bucketed, overlapped exchange inside train_step gives exactly the gradients of one
    all-reduce over the whole arena, the four ~25 MB buckets are launched in completion order, replicas stay identical."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_gpu_bf16_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    [p.join(timeout=120) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for r in range(world):
        assert got[r][2], "bucketed gradients differ from the single all-reduce"
        assert [b for b, _, _ in got[r][3]] == [0, 1, 2, 3]
        assert got[r][6], "DH_DDP_OVERLAP=0 (exchange after the backward pass) changed the gradients"
    assert np.array_equal(got[0][4], got[1][4]) and np.array_equal(got[0][5], got[1][5])


def _gpu_f32_bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deephisto_amd.models.patch_cls_simple.model import get_model
    from oracle import resnet18 as oracle_net
    dev = torch.device("cuda:0")
    ref = oracle_net.seeded_model(21, 5, perturb_bn=True)
    outs = []
    for bucket_bytes in (0, 8 * 1024 * 1024):      # one collective vs six buckets
        m = get_model(5, "f32")
        m.load_state_dict(ref.state_dict())
        m.to(dev).train()
        g = torch.Generator().manual_seed(300 + rank)
        x = torch.rand(4, 3, 64, 64, generator=g).to(dev)
        y = torch.randint(0, 5, (4,), generator=g).to(dev)
        m.train_step(x, y, lr=1e-4, bucket_bytes=bucket_bytes)
        outs.append((m.flat_gradients(dev).clone().cpu(), list(m.overlap_log)))
    q.put((rank, bool(torch.equal(outs[0][0], outs[1][0])), outs[0][1], outs[1][1], int(outs[0][0].numel())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_ddp_resnet18_f32_buckets_cover_arena_from_the_end(built_lib):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_gpu_f32_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    got = [q.get(timeout=600) for _ in range(world)]
    [p.join(timeout=120) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for _, same, log1, logn, n in got:
        assert same                                            # bucketing does not change a single bit of the averaged gradients
        assert len(log1) == 1 and log1[0][1:] == (0, n)
        assert len(logn) >= 4
        ends = [o + c for _, o, c in logn]
        assert ends[0] == n and logn[-1][1] == 0               # first bucket = tail of the arena (fc, layer4), last = stem
        assert all(logn[i][1] == ends[i + 1] for i in range(len(logn) - 1))
