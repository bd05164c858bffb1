"""GPU parity, model side (row a6): HIP ResNet-18 forward vs the torch-CPU oracle.

Tolerances (floating point, stated per BASELINE.json north_star):
  f32 compute: |logit - oracle| <= 1e-4 absolute;
  bf16 compute: <= 2e-2 * max(1, max|logit|)  (bf16 has 8 significant bits and the
  activations are re-rounded after each of 20 convs: sqrt(20) * 2^-9 ~ 0.9e-2 of the activation scale).
The oracle is "parity unpinned" against torchvision (see oracle/resnet18.py)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import resnet18 as oracle_net
from oracle import synth, tiling

pytestmark = pytest.mark.gpu

F32_ATOL = 1e-4


@pytest.fixture(scope="module")
def dev(built_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _hip_model(oracle, dev, dtype="f32"):
    from deephisto_amd.models.patch_cls_simple.model import get_model
    m = get_model(5, compute_dtype=dtype)
    missing, unexpected = m.load_state_dict(oracle.state_dict(), strict=True)
    return m.to(dev).eval()


def test_state_dict_keys_match_torchvision_layout():
    from deephisto_amd.models.patch_cls_simple.model import get_model
    a = get_model(5).state_dict()
    b = oracle_net.ResNet18Oracle(5).state_dict()
    assert list(a.keys()) == list(b.keys())
    assert all(a[k].shape == b[k].shape for k in a)
    assert "layer2.0.downsample.1.running_var" in a and a["fc.weight"].shape == (5, 512)
    assert sum(v.numel() for k, v in a.items() if "running" not in k and "tracked" not in k) == 11179077


@pytest.mark.parametrize("dtype,ks,stride,cin,cout,hw", [
    ("f32", 3, 1, 64, 64, 32), ("f32", 3, 2, 64, 128, 32), ("f32", 1, 2, 64, 128, 32),
    ("f32", 3, 1, 128, 128, 14), ("f32", 3, 1, 512, 512, 8), ("f32", 3, 1, 512, 512, 7),
    ("bf16", 3, 1, 64, 64, 32), ("bf16", 3, 2, 128, 256, 28), ("bf16", 1, 2, 256, 512, 16),
    ("bf16", 3, 1, 256, 256, 16),
])
def test_single_conv_layer(dev, dtype, ks, stride, cin, cout, hw):
    """One conv + scale/shift + residual + ReLU through dh_debug_conv_bn_act."""
    from deephisto_amd._lib import check, lib
    g = torch.Generator().manual_seed(ks * 1000 + cin + hw)
    B = 3
    x = torch.randn(B, cin, hw, hw, generator=g)
    w = torch.randn(cout, cin, ks, ks, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5
    sc = 0.5 + torch.rand(cout, generator=g)
    sh = 0.2 * torch.randn(cout, generator=g)
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    if dtype == "bf16":
        x, w = x.to(tdt).float(), w.to(tdt).float()
    y = F.conv2d(x, w, None, stride, ks // 2)
    res = torch.randn(y.shape, generator=g).to(tdt).float()
    want = F.relu(y * sc[None, :, None, None] + sh[None, :, None, None] + res)
    x_d = x.permute(0, 2, 3, 1).contiguous().to(dev, tdt)
    r_d = res.permute(0, 2, 3, 1).contiguous().to(dev, tdt)
    out = torch.empty((B, y.shape[2], y.shape[3], cout), dtype=tdt, device=dev)
    wc, scc, shc = w.contiguous(), sc.contiguous(), sh.contiguous()
    check(lib().dh_debug_conv_bn_act(x_d.data_ptr(), wc.data_ptr(), scc.data_ptr(), shc.data_ptr(), r_d.data_ptr(),
                                     out.data_ptr(), B, hw, hw, cin, cout, ks, stride, 1,
                                     0 if dtype == "f32" else 1, None), "dh_debug_conv_bn_act")
    got = out.float().cpu().permute(0, 3, 1, 2)
    tol = 2e-5 * max(1.0, float(want.abs().max())) if dtype == "f32" else 1e-2 * max(1.0, float(want.abs().max()))
    assert float((got - want).abs().max()) <= tol


@pytest.mark.parametrize("P,B", [(256, 4), (224, 3), (96, 2)])
def test_resnet18_f32_logits_within_1e4(dev, P, B):
    oracle = oracle_net.seeded_model(123, 5, perturb_bn=True).eval()
    model = _hip_model(oracle, dev, "f32")
    host = synth.synth_slide(P + 40, P * B + 17, seed=P)
    o = np.array([[7 + 3 * i, 5 + i * P] for i in range(B)], np.int32)
    x = torch.from_numpy(tiling.features_nchw_predictor(host, o, P))
    with torch.no_grad():
        want = oracle(x)
    got = model(x.to(dev))
    # stem first, to localise failures
    from deephisto_amd._lib import check, lib
    H1 = (P - 1) // 2 + 1
    stem = torch.empty((B, H1, H1, 64), dtype=torch.float32, device=dev)
    check(lib().dh_debug_stem_out(model._handle, B, P, stem.data_ptr(), None), "dh_debug_stem_out")
    with torch.no_grad():
        stem_want = F.relu(oracle.bn1(oracle.conv1(x)))
    err = float((stem.cpu().permute(0, 3, 1, 2) - stem_want).abs().max())
    assert err <= 2e-5 * max(1.0, float(stem_want.abs().max())), f"stem error {err}"
    err = float((got.cpu() - want).abs().max())
    assert err <= F32_ATOL, f"max |logit error| = {err} (logit scale {float(want.abs().max())})"
    # fused gather path gives the same logits (same kernels, pixels read from the uint8 slide)
    slide = torch.from_numpy(host).to(dev)
    got2 = model.forward_tiles(slide, torch.from_numpy(o).to(dev), P)
    assert float((got2 - got).abs().max()) <= 1e-6


def test_resnet18_f32_batch64_and_determinism(dev):
    """B=64 (4-image patches in layer4 all full) and run-to-run bitwise determinism."""
    oracle = oracle_net.seeded_model(5, 5, perturb_bn=True).eval()
    model = _hip_model(oracle, dev, "f32")
    g = torch.Generator().manual_seed(1)
    x = torch.rand(10, 3, 128, 128, generator=g)
    with torch.no_grad():
        want = oracle(x)
    a = model(x.to(dev))
    b = model(x.to(dev))
    assert torch.equal(a, b)
    assert float((a.cpu() - want).abs().max()) <= F32_ATOL
    # per-tile result does not depend on batch composition
    c = model(x[3:7].to(dev))
    assert torch.equal(c, a[3:7])


def test_resnet18_bf16_logits(dev):
    oracle = oracle_net.seeded_model(321, 5, perturb_bn=True).eval()
    model = _hip_model(oracle, dev, "bf16")
    g = torch.Generator().manual_seed(2)
    x = torch.rand(6, 3, 256, 256, generator=g)
    with torch.no_grad():
        want = oracle(x)
    got = model(x.to(dev)).cpu()
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= 2e-2 * scale, f"bf16 logit error {err} at scale {scale}"


def test_model_errors(dev):
    from deephisto_amd.models.patch_cls_simple.model import get_model
    out = get_model(5, "bf16").to(dev).train()(torch.rand(2, 3, 64, 64, device=dev))   # bf16 training: the dh_train2 engine
    assert out.shape == (2, 5) and out.requires_grad
    with pytest.raises(ValueError):
        get_model(5, arch="resnet34")
    with pytest.raises(Exception, match="P"):                                           # bf16 engine: 64 <= P <= 256
        get_model(5, arch="resnet50").to(dev).train()(torch.zeros(1, 3, 288, 288, device=dev))
    m = get_model(5).to(dev)
    m.eval()
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 4, 64, 64, device=dev))


@pytest.mark.parametrize("dtype,P,n", [("bf16", 256, 70), ("bf16", 224, 300), ("f32", 128, 130), ("bf16", 256, 4096), ("f32", 256, 1024),
                                         ("bf16", 224, 4096), ("f32", 224, 1024)])
def test_large_launch_matches_small_launches(dev, dtype, P, n):
    """Launches with >= 256 conv tiles use the XCD-grouped persistent schedule (several iterations per workgroup,
    the last one partial, resident weights, 512-pixel tiles); launches of a few tiles use one tile per workgroup and
    smaller tile variants.  The per-tile arithmetic is the same, so the logits must be IDENTICAL -- and the small
    launches are the ones checked against the CPU oracle above.  n = 4 096 at P = 256 in bf16 is the library's maximum: a
    64 x 64 x 64-channel map is exactly 2^31 bytes and the schedule tables carry byte offsets as 32-bit words (unsigned in the
    kernel; tests/test_conv_tables_host.py sweeps the host tables under sanitizers); n = 1 024 is the float32 maximum.
    P = 224 at n = 4 096 / 1 024 (round 5): the reference's own patch size at the launch size bench.py's `p224` object is timed at -- its
    14 x 14 and 7 x 7 maps run on the FIT tiles there (five 7 x 14 half-images / ten whole 7 x 7 images per 512 slots, shared zero halos,
    lanes dealt to pixels by the host: conv3_tables_host.h), the small launches on power-of-two tiles."""
    from deephisto_amd import tiles
    oracle = oracle_net.seeded_model(31, 5, perturb_bn=True).eval()
    model = _hip_model(oracle, dev, dtype)
    side = 4096
    slide = tiles.synth_slide(side, side, 3, dev)
    rng = np.random.default_rng(n)
    o = np.stack([rng.integers(0, side - P, n), rng.integers(0, side - P, n)], 1).astype(np.int32)
    o_dev = torch.from_numpy(o).to(dev)
    big = model.forward_tiles(slide, o_dev, P)
    if n > 512:   # compare a spread of the tiles (first, last, a stride through the middle) instead of 600 small launches
        pick = np.unique(np.concatenate([np.arange(0, 21), np.arange(n - 21, n), np.arange(0, n, max(1, n // 40))]))
        sel = torch.from_numpy(pick).to(dev)
        small = torch.cat([model.forward_tiles(slide, o_dev[sel[i:i + 7]].contiguous(), P) for i in range(0, len(pick), 7)])
        big = big[sel]
    else:
        small = torch.cat([model.forward_tiles(slide, o_dev[i:i + 7].contiguous(), P) for i in range(0, n, 7)])
    assert torch.equal(big, small)
    assert bool(torch.isfinite(big).all()) and float(big.abs().max()) > 0


@pytest.mark.parametrize("P,side", [(256, 700), (224, 700), (96, 300), (100, 300), (64, 64), (330, 700), (32, 90)])
def test_fused_bf16_stem_pool_every_pixel(dev, P, side):
    """The fused bf16 stem (conv 7x7/2 + BN + ReLU + maxpool 3x3/2, one persistent kernel reading the uint8 slide) against a
    torch-CPU restatement with the SAME roundings (pixels k/255 -> bf16, weights -> bf16, f32 accumulation, BN in f32, one
    rounding to bf16): every pooled pixel of every tile, tiles in the slide's corners included (the first and the last byte of
    the allocation lie inside their windows).  Only the f32 summation order differs, so values agree to one bf16 ulp and
    nearly all are identical."""
    from deephisto_amd._lib import check, lib
    oracle = oracle_net.seeded_model(77, 5, perturb_bn=True).eval()
    model = _hip_model(oracle, dev, "bf16")
    host = synth.synth_slide(side, side + 37 if side > P else side, seed=P)
    H, W = host.shape[:2]
    rng = np.random.default_rng(P)
    o = [[0, 0], [H - P, W - P], [0, W - P], [H - P, 0]]
    o += [[int(rng.integers(0, H - P + 1)), int(rng.integers(0, W - P + 1))] for _ in range(9)]
    o = np.array(o, np.int32)
    n = len(o)
    model(torch.zeros(1, 3, P, P, device=dev))                         # finalises the handle
    H2 = ((P - 1) // 2 + 1 - 1) // 2 + 1
    got = torch.empty((n, H2, H2, 64), dtype=torch.float32, device=dev)
    slide = torch.from_numpy(host).to(dev)
    check(lib().dh_debug_stem_pool_bf16(model._handle, slide.data_ptr(), H, W, torch.from_numpy(o).to(dev).data_ptr(), n, P,
                                        got.data_ptr(), None), "dh_debug_stem_pool_bf16")
    x = torch.from_numpy(tiling.features_nchw_predictor(host, o, P)).bfloat16().float()
    with torch.no_grad():
        z = F.conv2d(x, oracle.conv1.weight.bfloat16().float(), None, 2, 3)
        bn = oracle.bn1
        sc = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        sh = bn.bias - bn.running_mean * sc
        y = F.relu((z * sc[None, :, None, None] + sh[None, :, None, None]).bfloat16().float())
        want = F.max_pool2d(y, 3, 2, 1).permute(0, 2, 3, 1)
    g = got.cpu()
    assert g.shape == want.shape
    err = (g - want).abs()
    assert bool((err <= 2.0 ** -7 * want.abs().clamp_min(2.0 ** -6)).all()), f"max err {float(err.max())}"
    assert float((g == want).float().mean()) > 0.98


@pytest.mark.parametrize("P", [256, 224, 96])
def test_wide_stride2_kernel_matches_128_pixel_kernel(dev, P, tmp_path):
    """Round 4: the wide stride-2 + downsample kernel (128 couts x 256 pixels per workgroup on half-chunk stages, conv3x3.inc HALF; its
    input written in 16-channel planes by the conv before it, into the third activation buffer) against the 128-pixel kernel it
    replaces (DH_CONV_S2_WIDE=0, read once per process: two fresh processes).  Same products, another summation order: the logits of
    300 tiles agree within bf16 noise (measured 3.7e-3 / 4.2e-3 at |logit| <= 2), and both stay within the stated 2e-2 of the float32
    oracle.  Each process runs the launch three times and requires identical bits: the first version of the 16-channel-plane output
    wrote IN PLACE over its 32-channel-plane residual and corrupted the tiles at the iteration boundaries of the producing kernel,
    differently from run to run.  P = 224: ragged tiles (56 / 28 / 14-pixel maps); P = 96: only layer 2's stride-2 conv is wide."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    helper = Path(__file__).resolve().parent / "helpers" / "s2_forward.py"
    outs = {}
    for wide in ("1", "0"):
        f = tmp_path / f"logits_{wide}.npy"
        env = dict(os.environ, DH_CONV_S2_WIDE=wide)
        r = subprocess.run([sys.executable, str(helper), str(f), str(P)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs[wide] = np.load(f)
    scale = max(1.0, float(np.abs(outs["0"]).max()))
    assert float(np.abs(outs["1"] - outs["0"]).max()) <= 1e-2 * scale
    oracle = oracle_net.seeded_model(31, 5, perturb_bn=True).eval()
    host = synth.synth_slide(4096, 4096, 3)
    rng = np.random.default_rng(1)
    o = np.stack([rng.integers(0, 4096 - P, 300), rng.integers(0, 4096 - P, 300)], 1).astype(np.int32)
    sel = np.r_[0:8, 36, 73, 109, 146, 219, 292]          # the first tiles and the ones the aliasing bug hit
    with torch.no_grad():
        want = oracle(torch.from_numpy(tiling.features_nchw_predictor(host, o[sel], P))).numpy()
    for wide in ("1", "0"):
        assert float(np.abs(outs[wide][sel] - want).max()) <= 2e-2 * max(1.0, float(np.abs(want).max())), wide
