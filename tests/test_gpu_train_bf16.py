"""GPU parity of the bf16 training engine (dh_train2): ResNet-50 (BASELINE configs[4]) and ResNet-18 in bf16.

Oracles (both "parity unpinned" against torchvision, see oracle/resnet50.py):
  * oracle/resnet50.py / resnet18.py: torch-CPU float32 restatement -- the reference for the STATED bf16 tolerance
    end to end (logits, loss, per-conv activations);
  * oracle/bf16_emulation.py: the same network with bf16 rounding at the engine's storage points -- separates kernel
    errors from bf16 arithmetic (the first convolutions agree to 1e-6 relative, tools/t2_check.py) and, with the engine's
    own ReLU patterns imposed, checks every gradient tensor.

Stated tolerances (bf16 = 8 significand bits, 2^-9 relative rounding per stored activation, amplified layer by layer in
a randomly initialised network with batch-statistic BN -- measured values in brackets, ResNet-50 B=16 P=96):
  logits vs float32 oracle          <= 5e-2 absolute at |logit| <= 0.4   [1.8e-2]
  logits vs bf16-emulating oracle   <= 3e-2                              [6e-3 .. 9e-3]
  loss   vs float32 oracle          <= 1e-2                              [1e-4]
  conv outputs vs float32 oracle    <= 1e-1 relative L2 per conv         [<= 5e-2]
  gradients vs bf16-emulating oracle with the engine's ReLU patterns, stored activations, loss gradient and gradient rounding
  points imposed: relative L2 <= 2e-2 per tensor  [<= 1.5e-2; the stem's BN bias <= 5e-2, measured <= 3.6e-2: see GRAD_GATE_LOOSE]
The ResNet-50 cases damp every block's last BN gain to 0.2 (any parameter values are legitimate for an arithmetic check):
with gain ~1 a random-init 50-layer network amplifies bf16 rounding to tens of percent at the logits in BOTH the engine and
the bf16-emulating oracle (measured: 0.6 relative L2 at the last conv), which says nothing about the kernels."""
import copy
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import resnet18 as o18
from oracle import resnet50 as o50
from oracle.bf16_emulation import forward_bf16

pytestmark = pytest.mark.gpu

# Composition gate of the backward pass (round 4, VERDICT r3 item 5: was 8e-2): relative L2 per gradient tensor against the bf16-emulating
# oracle run with the ENGINE's ReLU patterns, the engine's stored activations as the operands of every layer (`forced`), the engine's loss
# gradient, and bf16 rounding of every activation gradient where the engine stores one (`grad_rounding`).  What is left is summation order and
# the exact position of a few roundings; it accumulates from the loss down: measured <= 1.5e-2 at layer 1 / the stem, <= 1e-2 above.
GRAD_GATE = 2e-2
# the stem's BN bias: its gradient is the plain sum of ALL masked gradient elements at the bottom of the network (0.4-3 M signed terms that
# cancel to a fraction of their norm), so the bf16 rounding of each term shows undamped: measured 1.2e-2 ... 3.6e-2 over the six cases
GRAD_GATE_LOOSE = {"bn1.bias": 5e-2}
# ... and the same comparison WITHOUT the forced activations: the emulation hands its own activations from layer to
# layer, so a hand-off the engine got wrong between two layers (a stale or foreign buffer as an operand) shows as a gradient error of order one
# in every tensor below it, where the forced run -- a chain of per-layer checks -- would not see it.  What is left here is the forward
# difference of two bf16 networks entering through the activation operands: measured <= 4.5e-2 per tensor, worst in layer 4 (round 3 gated this run at 8e-2).
GRAD_GATE_FREE = 6e-2


@pytest.fixture(scope="module")
def dev(built_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _pair(dev, arch, seed, gain):
    from deephisto_amd.models.patch_cls_simple.model import get_model
    onet = o50 if arch == "resnet50" else o18
    ref = onet.seeded_model(seed, 5, perturb_bn=True)
    if gain != 1.0:
        with torch.no_grad():
            for name, mod in ref.named_modules():
                if name.endswith("bn3" if arch == "resnet50" else "bn2"):
                    mod.weight.mul_(gain)
    m = get_model(5, "bf16", arch=arch)
    m.load_state_dict(ref.state_dict())
    return ref.train(), m.to(dev).train()


def _engine(m):
    return m._engine if hasattr(m, "_engine") else m._engine2


def _act(m, name, what, shape, dev):
    from deephisto_amd._lib import check, lib
    n = int(np.prod(shape))
    buf = torch.empty(n, dtype=torch.float32, device=dev)
    check(lib().dh_train2_debug_act(_engine(m).handle, name.encode(), what, buf.data_ptr(), n, None), "dh_train2_debug_act")
    b, c, h, w = shape
    return buf.cpu().reshape(b, h, w, c).permute(0, 3, 1, 2)


@pytest.mark.parametrize("arch,B,P,gain", [("resnet50", 16, 96, 0.2), ("resnet18", 16, 64, 1.0), ("resnet50", 6, 64, 0.2)])
def test_forward_backward_vs_oracles(dev, arch, B, P, gain):
    ref, m = _pair(dev, arch, 3, gain)
    g = torch.Generator().manual_seed(B * P)
    x = torch.rand(B, 3, P, P, generator=g)
    y = torch.randint(0, 5, (B,), generator=g)
    acts = {}
    for name, mod in ref.named_modules():
        if isinstance(mod, torch.nn.Conv2d):
            mod.register_forward_hook(lambda _m, _i, out, name=name: acts.__setitem__(name, out.detach()))
    emu = copy.deepcopy(ref)
    out_ref = ref(x)
    loss_ref = F.cross_entropy(out_ref, y)
    out = m(x.to(dev))
    loss = F.cross_entropy(out, y.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    got = out.detach().cpu()
    assert float((got - out_ref.detach()).abs().max()) <= 5e-2 * max(1.0, float(out_ref.abs().max()) / 0.4)
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-2
    for name, a in acts.items():
        z = _act(m, name, 0, a.shape, dev)
        assert float((z - a).norm() / a.norm()) <= 1e-1, name
    # the first convolutions see identical bf16 inputs in the engine and in the emulation: kernel-level agreement
    rec = {}
    out_emu = forward_bf16(copy.deepcopy(emu), x, rec)
    assert float((_act(m, "conv1", 0, rec["conv1"].shape, dev) - rec["conv1"]).norm() / rec["conv1"].norm()) <= 1e-4
    first = "layer1.0.conv1"
    assert float((_act(m, first, 0, rec[first].shape, dev) - rec[first]).norm() / rec[first].norm()) <= 2e-3
    assert float((got - out_emu.detach()).abs().max()) <= 3e-2
    # gradients: bf16-emulating oracle with the engine's ReLU patterns imposed
    masks = {name: _act(m, name, 1, a.shape, dev) > 0 for name, a in acts.items() if "downsample" not in name}
    emu.zero_grad()
    # ... and the ENGINE's loss gradient: dL/dlogits = (softmax(engine logits) - onehot) / B.  The two forwards differ by up to 3e-2 in
    # the logits, which alone moves every gradient by percents; with it imposed what is compared is the backward pass
    dl = (torch.softmax(got, 1) - F.one_hot(y, 5).float()) / B
    forced = {name: _act(m, name, 1, a.shape, dev) for name, a in acts.items() if "downsample" not in name}   # ... and its stored activations
    (forward_bf16(emu, x, None, masks, grad_rounding=True, forced=forced) * dl).sum().backward()
    want = {k: p.grad for k, p in emu.named_parameters()}
    bad, worst = {}, ("", 0.0)
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        e = float((p.grad.cpu() - want[k]).norm() / (want[k].norm() + 1e-30))
        if e > worst[1]:
            worst = (k, e)
        if e > GRAD_GATE_LOOSE.get(k, GRAD_GATE):
            bad[k] = e
    errs = sorted(((float((p.grad.cpu() - want[k]).norm() / (want[k].norm() + 1e-30)), k) for k, p in m.named_parameters()), reverse=True)
    print(f"[grad gate] {arch} B={B} P={P}: worst tensors " + ", ".join(f"{k} {e:.4f}" for e, k in errs[:5]))
    assert not bad, bad
    # free-running composition: the engine's ReLU patterns and loss gradient, the emulation's OWN activations between the layers
    emu_free = copy.deepcopy(ref)
    emu_free.zero_grad()
    (forward_bf16(emu_free, x, None, masks, grad_rounding=True) * dl).sum().backward()
    free = {k: p.grad for k, p in emu_free.named_parameters()}
    errs_free = sorted(((float((p.grad.cpu() - free[k]).norm() / (free[k].norm() + 1e-30)), k) for k, p in m.named_parameters()), reverse=True)
    print(f"[grad gate, free-running] {arch} B={B} P={P}: worst tensors " + ", ".join(f"{k} {e:.4f}" for e, k in errs_free[:5]))
    assert errs_free[0][0] <= GRAD_GATE_FREE, errs_free[:5]
    # running statistics and the batch counter went through
    sd, sr = m.state_dict(), ref.state_dict()
    for k in sr:
        if "running_mean" in k:
            assert float((sd[k].cpu() - sr[k]).abs().max()) <= 5e-2 * max(1.0, float(sr[k].abs().max())), k
        if "tracked" in k:
            assert int(sd[k]) == int(sr[k]) == 1


def test_resnet50_adam_steps_track_the_oracle(dev):
    """Fused HIP step (forward, CE, backward, Adam on f32 masters) for 4 steps on one batch vs torch Adam on the float32
    oracle: the loss trajectory stays within 2e-2 and decreases; state_dict round-trips into the oracle."""
    ref, m = _pair(dev, "resnet50", 7, 0.2)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(8, 3, 64, 64, generator=g)
    y = torch.randint(0, 5, (8,), generator=g)
    losses, want = [], []
    for _ in range(4):
        lr_, _ = o18.train_step(ref, opt, x, y)
        want.append(lr_)
        loss, logits = m.train_step(x.to(dev), y.to(dev), lr=1e-3)
        losses.append(float(loss))
    assert all(abs(a - b) <= 2e-2 for a, b in zip(losses, want)), (losses, want)
    assert losses[-1] < losses[0]
    sd = m.state_dict()
    assert set(sd) == set(ref.state_dict())
    probe = copy.deepcopy(ref)
    probe.load_state_dict(sd)          # keys / shapes interchange with the torchvision layout
    # evaluation mode (BN from the running statistics) against the oracle holding the SAME parameters
    xe = torch.rand(4, 3, 96, 96, generator=g)
    with torch.no_grad():
        we = probe.eval()(xe)
    ge = m.eval()(xe.to(dev)).cpu()
    assert float((ge - we).abs().max()) <= 5e-2 * max(1.0, float(we.abs().max()))


def test_torch_optimizer_loop_resnet18_bf16(dev):
    """The reference's loop shape on the bf16 engine: criterion + loss.backward() + torch.optim.Adam.step()."""
    ref, m = _pair(dev, "resnet18", 5, 1.0)
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-4)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(3)
    for step in range(3):
        x = torch.rand(16, 3, 64, 64, generator=g)
        y = torch.randint(0, 5, (16,), generator=g)
        l_ref, _ = o18.train_step(ref, opt_ref, x, y)
        opt.zero_grad()
        loss = F.cross_entropy(m(x.to(dev)), y.to(dev))
        loss.backward()
        opt.step()
        assert abs(float(loss) - l_ref) <= 2e-2, (step, float(loss), l_ref)
    # eval-mode inference afterwards uses the fast inference kernels with the updated parameters
    xe = torch.rand(4, 3, 96, 96, generator=g)
    with torch.no_grad():
        want = ref.eval()(xe)
    got = m.eval()(xe.to(dev)).cpu()
    assert float((got - want).abs().max()) <= 5e-2 * max(1.0, float(want.abs().max()))


def test_bf16_step_is_bit_reproducible(dev):
    outs = []
    g = torch.Generator().manual_seed(9)
    x = torch.rand(8, 3, 64, 64, generator=g).to(dev)
    y = torch.randint(0, 5, (8,), generator=g).to(dev)
    for _ in range(2):
        _, m = _pair(dev, "resnet50", 11, 0.5)
        l1, lg1 = m.train_step(x, y, lr=1e-3)
        grads = m.flat_gradients(dev).clone()
        l2, lg2 = m.train_step(x, y, lr=1e-3)
        outs.append((float(l1), lg1.clone(), grads, float(l2), lg2.clone()))
    a, b = outs
    assert a[0] == b[0] and a[3] == b[3] and torch.equal(a[1], b[1]) and torch.equal(a[4], b[4])
    assert torch.equal(a[2], b[2]), "gradient arena differs between two identical runs"


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_fused_backward_adam_equals_backward_then_adam(dev, arch):
    """dh_train2_backward_adam (each block's Adam update + bf16 repack behind its weight gradients on the side stream) against
    dh_train2_backward + dh_train2_adam_step: masters, moments-driven next steps and the repacked operands (seen through the
    next forward) must be identical bit for bit over several steps."""
    from deephisto_amd.models.patch_cls_simple.model import get_model
    g = torch.Generator().manual_seed(21)
    x = torch.rand(8, 3, 96, 96, generator=g).to(dev)
    y = torch.randint(0, 5, (8,), generator=g).to(dev)
    runs = []
    for fuse in (True, False):
        torch.manual_seed(3)
        m = get_model(5, "bf16", arch=arch).to(dev).train()
        eng = m._bf16_engine() if arch == "resnet18" else m._engine
        eng.fuse_optimizer = fuse
        trace = []
        for _ in range(4):
            loss, logits = m.train_step(x, y, lr=1e-3)
            trace.append((float(loss), logits.clone()))
        runs.append((trace, eng.flat(0, dev).clone()))
        del m
    (ta, pa), (tb, pb) = runs
    assert torch.equal(pa, pb), "parameter arenas differ"
    for (la, ga), (lb, gb) in zip(ta, tb):
        assert la == lb and torch.equal(ga, gb)


@pytest.mark.parametrize("arch,B,P", [("resnet18", 6, 96), ("resnet50", 4, 128)])
def test_bn_fold_is_bit_identical(dev, arch, B, P, monkeypatch):
    """Round 4: on small maps the BN apply passes add the partial rows of their own 64-channel slice in their
    prologue (bn_fold.inc) instead of waiting for a finalize launch.  The prologue adds the rows in the finalize kernel's own order,
    so an engine created with DH_T2_FOLD=0 (finalize launches everywhere) gives the same losses, logits, parameters and running
    statistics, bit for bit, over three optimiser steps."""
    from deephisto_amd.models.patch_cls_simple.model import get_model
    g = torch.Generator().manual_seed(47)
    x = torch.rand(B, 3, P, P, generator=g).to(dev)
    y = torch.randint(0, 5, (B,), generator=g).to(dev)
    runs = []
    for fold in ("1", "0"):
        monkeypatch.setenv("DH_T2_FOLD", fold)
        torch.manual_seed(5)
        m = get_model(5, "bf16", arch=arch).to(dev).train()
        trace = []
        for _ in range(3):
            loss, logits = m.train_step(x, y, lr=1e-3)
            trace.append((float(loss), logits.clone()))
        eng = m._bf16_engine() if arch == "resnet18" else m._engine
        runs.append((trace, eng.flat(0, dev).clone(), eng.flat(2, dev).clone()))
        del m
    (ta, pa, ra), (tb, pb, rb) = runs
    for (la, ga), (lb, gb) in zip(ta, tb):
        assert la == lb and torch.equal(ga, gb)
    assert torch.equal(pa, pb) and torch.equal(ra, rb)
    assert bool(torch.isfinite(pa).all())


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_join_bn_fusion_is_bit_identical(dev, arch, monkeypatch):
    """The downsample branch's BN applied inside the join BN's pass (bn2_apply_join_kernel: the branch value rounded to bf16 where the
    separate pass would have stored it) against the two-pass form (DH_T2_JOIN=0, read when the engine is created): identical bits."""
    from deephisto_amd.models.patch_cls_simple.model import get_model
    g = torch.Generator().manual_seed(31)
    x = torch.rand(6, 3, 96, 96, generator=g).to(dev)
    y = torch.randint(0, 5, (6,), generator=g).to(dev)
    runs = []
    for join in ("1", "0"):
        monkeypatch.setenv("DH_T2_JOIN", join)
        torch.manual_seed(5)
        m = get_model(5, "bf16", arch=arch).to(dev).train()
        trace = []
        for _ in range(3):
            loss, logits = m.train_step(x, y, lr=1e-3)
            trace.append((float(loss), logits.clone()))
        eng = m._bf16_engine() if arch == "resnet18" else m._engine
        runs.append((trace, eng.flat(0, dev).clone()))
        del m
    (ta, pa), (tb, pb) = runs
    for (la, ga), (lb, gb) in zip(ta, tb):
        assert la == lb and torch.equal(ga, gb)
    assert torch.equal(pa, pb)


def test_bench_shape_runs_and_learns(dev):
    """64 x 224^2 (BASELINE configs[4] per-rank batch): every layer takes its large-launch path; the loss on a fixed batch
    must fall (no CPU oracle at this size: ResNet-50 forward + backward of 64 images is minutes of host time)."""
    from deephisto_amd.models.patch_cls_simple.model import get_model
    torch.manual_seed(0)
    m = get_model(5, arch="resnet50").to(dev).train()
    g = torch.Generator().manual_seed(1)
    x = torch.rand(64, 3, 224, 224, generator=g).to(dev)
    y = torch.randint(0, 5, (64,), generator=g).to(dev)
    losses = [float(m.train_step(x, y, lr=1e-3)[0]) for _ in range(6)]
    assert all(np.isfinite(losses)) and losses[-1] < 0.7 * losses[0], losses


def test_bench_shape_slices_against_the_emulation(dev):
    """64 x 224^2 (the bench shape: every layer on its large-launch path, fused GEMM epilogues, side-stream weight gradients) held
    against the bf16 emulation where that is affordable on the host: on the first 8 images of the batch
      * conv1's output Z (per-image work: the emulation's 7x7 convolution of the same bf16 inputs): 1e-4 relative L2;
      * layer1.0.conv1's output Z: the emulation's max-pool + 1x1 convolution applied to the ENGINE's normalised stem map of those
        images (the batch statistics come from all 64 images, so the slice takes them from the engine): 2e-3;
    and, on the whole batch, the fc gradients recomputed on the host from the engine's own logits and its last block output
    (dW = dlogits^T . pooled, db = sum dlogits; float64): 1e-3 relative (pooled features are bf16 activations averaged in float32)."""
    ref, m = _pair(dev, "resnet50", 4, 0.2)
    g = torch.Generator().manual_seed(2)
    B, P, S = 64, 224, 8
    x = torch.rand(B, 3, P, P, generator=g)
    y = torch.randint(0, 5, (B,), generator=g)
    loss, logits = m.train_step(x.to(dev), y.to(dev), lr=1e-4)
    grads = m.flat_gradients(dev).clone()
    assert np.isfinite(float(loss))
    from oracle.bf16_emulation import rb
    # conv1 Z of the first S images
    z1 = _act(m, "conv1", 0, (B, 64, 112, 112), dev)[:S]
    want1 = rb(F.conv2d(rb(x[:S]), rb(ref.conv1.weight.detach()), None, 2, 3))
    assert float((z1 - want1).norm() / want1.norm()) <= 1e-4
    # layer1.0.conv1 Z from the engine's normalised stem map
    y0 = _act(m, "conv1", 1, (B, 64, 112, 112), dev)[:S]
    want2 = rb(F.conv2d(F.max_pool2d(y0, 3, 2, 1), rb(ref.layer1[0].conv1.weight.detach())))
    z2 = _act(m, "layer1.0.conv1", 0, (B, 64, 56, 56), dev)[:S]
    assert float((z2 - want2).norm() / want2.norm()) <= 2e-3
    # fc gradients of the whole batch from the engine's logits and last block output
    ylast = _act(m, "layer4.2.conv3", 1, (B, 2048, 7, 7), dev).double()
    pooled = ylast.mean((2, 3))
    dl = (torch.softmax(logits.cpu().double(), 1) - F.one_hot(y, 5).double()) / B
    eng = _engine(m)
    off = {}
    from deephisto_amd._lib import check, lib
    gw = torch.empty(5, 2048, dtype=torch.float32, device=dev)
    gb = torch.empty(5, dtype=torch.float32, device=dev)
    check(lib().dh_train2_tensor(eng.handle, b"fc.weight", 1, gw.data_ptr(), gw.numel(), 0, None), "grad fc.weight")
    check(lib().dh_train2_tensor(eng.handle, b"fc.bias", 1, gb.data_ptr(), gb.numel(), 0, None), "grad fc.bias")
    torch.cuda.synchronize()
    want_w, want_b = dl.T @ pooled, dl.sum(0)
    assert float((gw.cpu().double() - want_w).norm() / want_w.norm()) <= 1e-3
    assert float((gb.cpu().double() - want_b).norm() / want_b.norm()) <= 1e-5
    assert grads.numel() == m.flat_gradients(dev).numel()


def test_bucket_layout_and_callback_order(dev):
    from deephisto_amd._lib import BUCKET_CB, check, lib
    _, m = _pair(dev, "resnet50", 2, 0.2)
    x = torch.rand(4, 3, 64, 64).to(dev)
    out = m(x)
    eng = _engine(m)
    n_total = m.flat_gradients(dev).numel()
    ranges = eng.bucket_ranges(25 * 1024 * 1024)
    assert len(ranges) == 4                                     # 94 MB of float32 gradients in ~25 MB buckets
    assert ranges[0][0] == 0 and sum(c for _, c in ranges) == n_total
    assert all(ranges[i][0] + ranges[i][1] == ranges[i + 1][0] for i in range(len(ranges) - 1))
    assert all(c * 4 >= 25 * 1024 * 1024 for _, c in ranges[:-1])
    seen = []
    cb = BUCKET_CB(lambda b, off, cnt, _u: seen.append((b, off, cnt)))
    check(lib().dh_train2_set_buckets(eng.handle, 25 * 1024 * 1024, cb, None, None), "set_buckets")
    dl = torch.zeros_like(out)
    check(lib().dh_train2_backward(eng.handle, dl.data_ptr(), None), "backward")
    check(lib().dh_train2_set_buckets(eng.handle, 0, None, None, None), "set_buckets")
    assert [s[0] for s in seen] == [0, 1, 2, 3] and [(o, c) for _, o, c in seen] == ranges
    # fc sits in the first bucket, the stem in the last: completion order of the backward pass
    sd_first = {k for k, _ in m.named_parameters()}
    assert "fc.weight" in sd_first


@pytest.mark.parametrize("arch,B,P", [("resnet50", 2, 256), ("resnet50", 3, 128), ("resnet18", 5, 224), ("resnet50", 1, 160)])
def test_shape_sweep_forward_backward(dev, arch, B, P):
    """Other patch sizes / batch sizes (odd maps: 160 -> 5x5, 224 -> 7x7; 256 -> the 64-pixel-wide wgrad rows; batch 1): forward
    against the bf16-emulating oracle, gradients finite and -- for the last layers, whose gradient does not pass through deep
    ReLU patterns -- close to the emulation's."""
    ref, m = _pair(dev, arch, 17, 0.2 if arch == "resnet50" else 1.0)
    g = torch.Generator().manual_seed(B + P)
    x = torch.rand(B, 3, P, P, generator=g)
    y = torch.randint(0, 5, (B,), generator=g)
    if B == 1:
        ref.eval(); m.eval()     # one sample per channel at the deepest maps has no batch statistics worth comparing
        with torch.no_grad():
            want = forward_bf16(ref, x)
            got = m(x.to(dev)).cpu()
        assert float((got - want).abs().max()) <= 3e-2 * max(1.0, float(want.abs().max()))
        return
    emu = copy.deepcopy(ref)
    out_emu = forward_bf16(emu, x)
    F.cross_entropy(out_emu, y).backward()
    out = m(x.to(dev))
    F.cross_entropy(out, y.to(dev)).backward()
    assert float((out.detach().cpu() - out_emu.detach()).abs().max()) <= 5e-2 * max(1.0, float(out_emu.detach().abs().max()))
    want = dict(emu.named_parameters())
    for k, p in m.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
    for k in ("fc.weight", "fc.bias"):
        e = float((dict(m.named_parameters())[k].grad.cpu() - want[k].grad).norm() / (want[k].grad.norm() + 1e-30))
        assert e <= 0.15, (k, e)
    # every gradient tensor against the emulation with the ENGINE's ReLU patterns, activations and loss gradient imposed (the 2e-2 gate of
    # test_forward_backward_vs_oracles, here at the sweep's shapes: 7 x 7 and 5 x 5 maps, 64-pixel-wide wgrad rows)
    shapes = {}
    hooks = [mod.register_forward_hook(lambda _m, _i, o, name=name: shapes.__setitem__(name, tuple(o.shape)))
             for name, mod in ref.named_modules() if isinstance(mod, torch.nn.Conv2d)]
    with torch.no_grad():
        ref(x)
    for h in hooks:
        h.remove()
    masks = {name: _act(m, name, 1, shp, dev) > 0 for name, shp in shapes.items() if "downsample" not in name}
    emu2 = copy.deepcopy(ref)
    emu2.zero_grad()
    dl = (torch.softmax(out.detach().cpu(), 1) - F.one_hot(y, 5).float()) / B     # the engine's own loss gradient (see test_forward_backward_vs_oracles)
    forced = {name: _act(m, name, 1, shp, dev) for name, shp in shapes.items() if "downsample" not in name}
    (forward_bf16(emu2, x, None, masks, grad_rounding=True, forced=forced) * dl).sum().backward()
    want2 = {k: p.grad for k, p in emu2.named_parameters()}
    bad, worst = {}, ("", 0.0)
    for k, p in m.named_parameters():
        e = float((p.grad.cpu() - want2[k]).norm() / (want2[k].norm() + 1e-30))
        if e > worst[1]:
            worst = (k, e)
        if e > GRAD_GATE_LOOSE.get(k, GRAD_GATE):
            bad[k] = e
    errs = sorted(((float((p.grad.cpu() - want2[k]).norm() / (want2[k].norm() + 1e-30)), k) for k, p in m.named_parameters()), reverse=True)
    print(f"[grad gate] sweep {arch} B={B} P={P}: worst tensors " + ", ".join(f"{k} {e:.4f}" for e, k in errs[:5]))
    assert not bad, bad
