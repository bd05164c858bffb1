"""GPU parity, row a7: HIP training step (forward with batch-stat BN, backward, Adam) vs the
torch-CPU oracle (oracle/resnet18.py, "parity unpinned" against torchvision).

Tolerances (float32 compute, stated per SURVEY section 8d): logits <= 1e-4 abs; loss <= 1e-4
after step 1 and <= 1e-3 after 3 steps; gradients: relative L2 error <= 2e-2 per tensor against a
float64 run of the oracle (see the comment in test_forward_backward_matches_oracle)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import resnet18 as oracle_net

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(built_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _pair(dev, seed):
    from deephisto_amd.models.patch_cls_simple.model import get_model
    ref = oracle_net.seeded_model(seed, 5, perturb_bn=True)
    m = get_model(5, "f32")
    m.load_state_dict(ref.state_dict())
    return ref.train(), m.to(dev).train()


def _rel(a, b):
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)


@pytest.mark.parametrize("B,P", [(8, 64), (4, 128), (3, 224)])
def test_forward_backward_matches_oracle(dev, B, P):
    ref, m = _pair(dev, 11)
    g = torch.Generator().manual_seed(B * P)
    x = torch.rand(B, 3, P, P, generator=g)
    y = torch.randint(0, 5, (B,), generator=g)
    out_ref = ref(x)
    loss_ref = F.cross_entropy(out_ref, y)
    loss_ref.backward()
    out = m(x.to(dev))
    loss = F.cross_entropy(out, y.to(dev))
    loss.backward()
    assert float((out.detach().cpu() - out_ref.detach()).abs().max()) <= 1e-4
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-4
    # running statistics (momentum 0.1, unbiased variance) and the batch counter
    sd, sr = m.state_dict(), ref.state_dict()
    for k in sr:
        if "running" in k:
            assert _rel(sd[k].cpu(), sr[k]) <= 1e-5, k
        if "tracked" in k:
            assert int(sd[k]) == int(sr[k]) == 1
    # Gradients against a float64 run of the oracle.  A ReLU whose pre-activation is within float32 rounding of
    # zero (|pre| ~ 1e-6; about one element per 10^5-element tensor) may switch differently in two float32
    # implementations, and with a handful of images one switched element moves a max-norm by percents -- the
    # float32 CPU oracle itself shows 1e-2 max-norm differences from float64 on these inputs (tools/grad_check.py).
    # A switch high in the network then perturbs every gradient below it (relative L2 of a few 1e-3).  The per-tensor
    # relative L2 error is robust to that and still exposes any real kernel error (O(0.1-1)).
    ref64 = copy.deepcopy(ref).double()
    ref64.zero_grad()
    F.cross_entropy(ref64(x.double()), y).backward()
    g64 = {k: p.grad for k, p in ref64.named_parameters()}
    worst = {}
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        worst[k] = float((p.grad.cpu().double() - g64[k]).norm()) / (float(g64[k].norm()) + 1e-300)
    bad = {k: v for k, v in worst.items() if v > 2e-2}
    assert not bad, bad

def test_three_adam_steps_torch_optimizer(dev):
    """The reference's loop shape: criterion + loss.backward() + torch.optim.Adam.step()."""
    ref, m = _pair(dev, 5)
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-4)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(3)
    for step in range(3):
        x = torch.rand(8, 3, 64, 64, generator=g)
        y = torch.randint(0, 5, (8,), generator=g)
        l_ref, _ = oracle_net.train_step(ref, opt_ref, x, y)
        opt.zero_grad()
        out = m(x.to(dev))
        loss = F.cross_entropy(out, y.to(dev))
        loss.backward()
        opt.step()
        assert abs(float(loss) - l_ref) <= (1e-4 if step == 0 else 1e-3), (step, float(loss), l_ref)
    # Adam's update is ~lr*sign(g) while v is young: an element whose gradient is within rounding of
    # zero may move the other way, so single elements can differ by up to 2*lr per step; the bulk must agree.
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        d = (p.detach().cpu() - q.detach()).abs()
        assert float(d.max()) <= 6e-4 and float(d.mean()) <= 2e-5, (k, float(d.max()), float(d.mean()))


def test_fused_train_step_and_eval_roundtrip(dev):
    """HIP CrossEntropy + fused Adam (model.train_step), then eval-mode inference and a
    state_dict round trip through the oracle."""
    ref, m = _pair(dev, 9)
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(4)
    for step in range(3):
        x = torch.rand(8, 3, 64, 64, generator=g)
        y = torch.randint(0, 5, (8,), generator=g)
        l_ref, _ = oracle_net.train_step(ref, opt_ref, x, y)
        loss, _ = m.train_step(x.to(dev), y.to(dev), lr=1e-4)
        assert abs(float(loss) - l_ref) <= (1e-4 if step == 0 else 1e-3), (step, float(loss), l_ref)
    sd = m.state_dict()
    for k, v in ref.state_dict().items():
        if "tracked" in k:
            assert int(sd[k]) == int(v)
        elif "running" in k:
            assert _rel(sd[k].cpu(), v) <= 1e-4, k
        else:
            d = (sd[k].cpu() - v).abs()
            assert float(d.max()) <= 6e-4 and float(d.mean()) <= 2e-5, (k, float(d.max()), float(d.mean()))
    xe = torch.rand(4, 3, 96, 96, generator=g)
    with torch.no_grad():
        want = ref.eval()(xe)
    got = m.eval()(xe.to(dev))
    assert float((got.cpu() - want).abs().max()) <= 5e-4


def test_ce_loss_kernel(dev):
    from deephisto_amd._lib import check, lib
    g = torch.Generator().manual_seed(0)
    logits = (torch.randn(37, 5, generator=g) * 3).to(dev)
    y = torch.randint(0, 5, (37,), generator=g).to(dev)
    loss = torch.empty((), device=dev)
    dl = torch.empty_like(logits)
    check(lib().dh_ce_loss(logits.data_ptr(), y.data_ptr(), 37, 5, loss.data_ptr(), dl.data_ptr(), None), "ce")
    torch.cuda.synchronize()
    lr = logits.detach().cpu().requires_grad_(True)
    want = F.cross_entropy(lr, y.cpu())
    want.backward()
    assert abs(float(loss) - float(want)) <= 1e-5
    assert float((dl.cpu() - lr.grad).abs().max()) <= 1e-6


def test_bench_shape_step_matches_oracle(dev):
    """Batch 64 x 224^2 (the bench's training shape): the convolutions run with >= 256 tiles per launch, i.e. the
    persistent XCD-grouped schedule, the 9-tap wgrad with 14-row slabs, the 8x64 / 16x16 / 8x8x2 tile variants --
    none of which the small parity cases above reach.  Loss and per-tensor relative L2 of the gradients against the
    float32 CPU oracle (ReLU-mask switches allowed for, see test_forward_backward_matches_oracle)."""
    ref, m = _pair(dev, 5)
    g = torch.Generator().manual_seed(64224)
    x = torch.rand(64, 3, 224, 224, generator=g)
    y = torch.randint(0, 5, (64,), generator=g)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    out_ref = ref(x)
    loss_ref = F.cross_entropy(out_ref, y)
    loss_ref.backward()
    out = m(x.to(dev))
    loss = F.cross_entropy(out, y.to(dev))
    loss.backward()
    assert float((out.detach().cpu() - out_ref.detach()).abs().max()) <= 1e-4
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-4
    ref_g = {k: p.grad for k, p in ref.named_parameters()}
    bad = {}
    for k, p in m.named_parameters():
        e = float((p.grad.cpu() - ref_g[k]).norm()) / (float(ref_g[k].norm()) + 1e-30)
        if e > 2e-2:
            bad[k] = e
    assert not bad, bad


def test_batch_shape_change_keeps_optimizer_state(dev):
    """a change of the batch shape inside a train_step loop (short last batch) must keep the Adam
    moments and step count.  B = 8, then 4, then 8 against torch.optim.Adam on the oracle."""
    ref, m = _pair(dev, 13)
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(8)
    for step, B in enumerate([8, 4, 8, 4]):
        x = torch.rand(B, 3, 64, 64, generator=g)
        y = torch.randint(0, 5, (B,), generator=g)
        l_ref, _ = oracle_net.train_step(ref, opt_ref, x, y)
        loss, _ = m.train_step(x.to(dev), y.to(dev), lr=1e-4)
        assert abs(float(loss) - l_ref) <= (1e-4 if step == 0 else 1e-3), (step, float(loss), l_ref)
    sd = m.state_dict()
    for k, v in ref.state_dict().items():
        if "tracked" in k or "running" in k:
            continue
        d = (sd[k].cpu() - v).abs()
        # a restarted optimizer (moments zeroed, bias correction of step 3) would move every element by ~3 lr = 3e-4
        assert float(d.max()) <= 8e-4 and float(d.mean()) <= 2.5e-5, (k, float(d.max()), float(d.mean()))


def test_train_step_is_bit_reproducible(dev):
    """No float atomics anywhere in the step (3x3, 1x1 and stem wgrads go through slab buffers summed in a fixed
    order): two runs from identical state give bit-identical gradients, logits and updated parameters."""
    from deephisto_amd.models.patch_cls_simple.model import get_model
    ref = oracle_net.seeded_model(21, 5, perturb_bn=True)
    g = torch.Generator().manual_seed(77)
    x = torch.rand(8, 3, 96, 96, generator=g).to(dev)
    y = torch.randint(0, 5, (8,), generator=g).to(dev)
    outs = []
    for _ in range(2):
        m = get_model(5, "f32")
        m.load_state_dict(ref.state_dict())
        m.to(dev).train()
        loss, logits = m.train_step(x, y, lr=1e-3)
        grads = m.flat_gradients(dev).clone()
        loss2, logits2 = m.train_step(x, y, lr=1e-3)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        outs.append((float(loss), logits.clone(), grads, float(loss2), logits2.clone(), sd))
    a, b = outs
    assert a[0] == b[0] and a[3] == b[3]
    assert torch.equal(a[1], b[1]) and torch.equal(a[4], b[4])
    assert torch.equal(a[2], b[2]), "gradient arena differs between two identical runs"
    for k in a[5]:
        assert torch.equal(a[5][k], b[5][k]), k


def test_fused_backward_adam_equals_backward_then_adam(dev):
    """dh_resnet18_backward_adam (each block's Adam update + repack behind its weight gradients on the side stream) against
    dh_resnet18_backward + dh_resnet18_adam_step: losses, logits of every step and the final state bit for bit."""
    from deephisto_amd.models.patch_cls_simple.model import get_model
    ref = oracle_net.seeded_model(22, 5, perturb_bn=True)
    g = torch.Generator().manual_seed(78)
    x = torch.rand(8, 3, 96, 96, generator=g).to(dev)
    y = torch.randint(0, 5, (8,), generator=g).to(dev)
    outs = []
    for fuse in (True, False):
        m = get_model(5, "f32")
        m.load_state_dict(ref.state_dict())
        m.to(dev).train()
        m.fuse_optimizer = fuse
        trace = [m.train_step(x, y, lr=1e-3) for _ in range(4)]
        trace = [(float(l), lg.clone()) for l, lg in trace]
        outs.append((trace, {k: v.clone() for k, v in m.state_dict().items()}))
    (ta, sa), (tb, sb) = outs
    for (la, ga), (lb, gb) in zip(ta, tb):
        assert la == lb and torch.equal(ga, gb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_ce_loss_rejects_out_of_range_label(dev):
    from deephisto_amd.models.patch_cls_simple.model import ce_loss
    logits = torch.randn(6, 5, device=dev)
    y = torch.tensor([0, 1, 7, 2, -1, 4], device=dev)
    loss, dl = ce_loss(logits, y, want_grad=True)
    assert torch.isnan(loss)                                   # torch raises here; the kernel flags it in the value
    assert float(dl[2].abs().max()) == 0.0 and float(dl[4].abs().max()) == 0.0 and float(dl[0].abs().max()) > 0.0
