"""Per-kernel parity of the float32 training engine's backward building blocks (VERDICT r1 #3 iii).

Every kernel that dh_resnet18_backward launches is run alone through its debug hook (include/deephisto_hip.h, "debug")
on random data and held against torch autograd on the CPU in float64: relative L2 error <= 1e-5 per output tensor
(float32 MFMA products are exact, sums of up to ~10^4 terms in float32: ~sqrt(K) * 6e-8).  The whole-network tests
(tests/test_gpu_train.py) can only afford 2e-2 because ReLU patterns flip between two float32 implementations; these
tests cannot hide a 1 % error in a BN coefficient or a wgrad edge tap."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def dev(built_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(got, want):
    want = want.double()
    return float((got.double() - want).norm() / (want.norm() + 1e-300))


def _nhwc(t, dev):
    return t.permute(0, 2, 3, 1).contiguous().to(dev, torch.float32)


def _nchw(t):
    return t.cpu().permute(0, 3, 1, 2)


@pytest.mark.parametrize("ks,stride,cin,cout,B,H,mode", [
    (3, 1, 64, 64, 3, 56, 0),      # fused 9-tap kernel, 1 row per step
    (3, 1, 128, 128, 4, 28, 0),    # 2 rows per step
    (3, 2, 64, 128, 3, 56, 0),     # stride 2 window
    (3, 1, 512, 512, 5, 7, 0),     # odd map, several images per slab
    (3, 1, 256, 256, 3, 14, 1),    # per-tap kernel forced (the path of rows that do not fit the fused plan)
    (1, 2, 64, 128, 3, 56, 0),     # downsample branch
    (1, 2, 256, 512, 4, 14, 0),
    (3, 1, 64, 64, 2, 72, 0),      # Wo > 64: falls back to the per-tap kernel by itself
])
def test_wgrad_kernels(dev, ks, stride, cin, cout, B, H, mode):
    from deephisto_amd._lib import check, lib
    g = torch.Generator().manual_seed(ks * 100 + cin + H)
    x = torch.randn(B, cin, H, H, generator=g, dtype=torch.float64)
    w = torch.zeros(cout, cin, ks, ks, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, None, stride, ks // 2)
    dz = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (want,) = torch.autograd.grad(y, w, dz)
    dw = torch.empty(cout, cin, ks, ks, dtype=torch.float32, device=dev)
    xd, dzd = _nhwc(x, dev), _nhwc(dz, dev)
    check(lib().dh_debug_wgrad_f32(dzd.data_ptr(), xd.data_ptr(), dw.data_ptr(), B, H, H, cin, cout, ks, stride, mode, None), "wgrad")
    assert _rel(dw.cpu(), want) <= TOL
    # edge taps on their own: a wrong border would drown in the full-tensor norm of a large map
    if ks == 3:
        for t in ((0, 0), (0, 2), (2, 0), (2, 2)):
            assert _rel(dw.cpu()[:, :, t[0], t[1]], want[:, :, t[0], t[1]]) <= TOL, t


@pytest.mark.parametrize("B,P", [(3, 64), (2, 224)])
def test_stem_wgrad_kernel(dev, B, P):
    from deephisto_amd._lib import check, lib
    g = torch.Generator().manual_seed(P)
    x = torch.rand(B, 3, P, P, generator=g, dtype=torch.float64)
    w = torch.zeros(64, 3, 7, 7, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, None, 2, 3)
    dz = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (want,) = torch.autograd.grad(y, w, dz)
    dw = torch.empty(64, 3, 7, 7, dtype=torch.float32, device=dev)
    xd = x.to(dev, torch.float32).contiguous()
    check(lib().dh_debug_stem_wgrad_f32(_nhwc(dz, dev).data_ptr(), xd.data_ptr(), dw.data_ptr(), B, P, None), "stem wgrad")
    assert _rel(dw.cpu(), want) <= TOL


@pytest.mark.parametrize("ks,stride,cin,cout,B,H,res", [
    (3, 1, 64, 64, 2, 32, True), (3, 1, 256, 256, 3, 14, False), (3, 2, 64, 128, 2, 32, False),
    (3, 2, 256, 512, 3, 14, False), (1, 2, 64, 128, 2, 32, True), (1, 2, 256, 512, 3, 14, True),
])
def test_dgrad_paths(dev, ks, stride, cin, cout, B, H, res):
    from deephisto_amd._lib import check, lib
    g = torch.Generator().manual_seed(ks * 10 + stride + cin)
    x = torch.zeros(B, cin, H, H, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cout, cin, ks, ks, generator=g, dtype=torch.float64) * (2.0 / (cin * ks * ks)) ** 0.5
    y = F.conv2d(x, w, None, stride, ks // 2)
    dz = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (want,) = torch.autograd.grad(y, x, dz)
    r = torch.randn(B, cin, H, H, generator=g, dtype=torch.float64) if res else None
    if res:
        want = want + r
    dx = torch.empty(B, H, H, cin, dtype=torch.float32, device=dev)
    wd = w.to(dev, torch.float32).contiguous()
    rd = _nhwc(r, dev) if res else None
    check(lib().dh_debug_dgrad_f32(_nhwc(dz, dev).data_ptr(), wd.data_ptr(), rd.data_ptr() if res else None, dx.data_ptr(),
                                   B, H, H, cin, cout, ks, stride, None), "dgrad")
    assert _rel(_nchw(dx), want) <= TOL


@pytest.mark.parametrize("C,B,H,relu,res", [(64, 4, 28, True, False), (128, 3, 14, True, True), (512, 6, 7, False, False),
                                            (256, 2, 56, True, True)])
def test_bn_forward_backward_kernels(dev, C, B, H, relu, res):
    from deephisto_amd._lib import check, lib
    g = torch.Generator().manual_seed(C + H)
    z = (torch.randn(B, C, H, H, generator=g, dtype=torch.float64) * 1.7 + 0.4).requires_grad_(True)
    gamma = (0.5 + torch.rand(C, generator=g, dtype=torch.float64)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g, dtype=torch.float64)).requires_grad_(True)
    r = torch.randn(B, C, H, H, generator=g, dtype=torch.float64) if res else None
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    pre = F.batch_norm(z, rm, rv, gamma, beta, True, 0.1, 1e-5)
    if res:
        pre = pre + r
    y = F.relu(pre) if relu else pre
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    dz_w, dg_w, db_w = torch.autograd.grad(y, (z, gamma, beta), dy)
    g_w = dy * (y > 0) if relu else dy
    rows = B * H * H
    zd, dyd = _nhwc(z.detach(), dev), _nhwc(dy, dev)
    yd, dzd, gd = torch.empty_like(zd), torch.empty_like(zd), torch.empty_like(zd)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    st = torch.empty(4 * C, device=dev)
    gmd, btd = gamma.detach().to(dev, torch.float32), beta.detach().to(dev, torch.float32)
    rd = _nhwc(r, dev) if res else None
    check(lib().dh_debug_bn_f32(zd.data_ptr(), gmd.data_ptr(), btd.data_ptr(), rd.data_ptr() if res else None, 1 if relu else 0,
                                yd.data_ptr(), dyd.data_ptr(), dzd.data_ptr(), gd.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                st.data_ptr(), rows, C, None), "bn")
    assert _rel(_nchw(yd), y.detach()) <= TOL
    # the ReLU pattern of the float32 forward can differ from float64 where |pre-activation| ~ 1e-7: compare the backward
    # with torch's pattern imposed on the few such elements excluded (none in practice at these sizes)
    same_mask = (_nchw(yd) > 0) == (y.detach() > 0) if relu else torch.ones_like(y, dtype=torch.bool)
    assert float(same_mask.double().mean()) >= 0.99999
    assert _rel(_nchw(gd), g_w) <= TOL
    assert _rel(dg.cpu(), dg_w) <= TOL and _rel(db.cpu(), db_w) <= TOL
    assert _rel(_nchw(dzd), dz_w) <= 2 * TOL      # dz = k0 g + k1 z + k2: three rounded coefficients
    s = st.cpu().double()
    mean_w = z.detach().mean((0, 2, 3))
    var_w = z.detach().var((0, 2, 3), unbiased=False)
    assert _rel(s[:C], mean_w) <= TOL and _rel(s[C:2 * C], 1.0 / torch.sqrt(var_w + 1e-5)) <= TOL
    assert _rel(s[2 * C:3 * C], rm) <= TOL and _rel(s[3 * C:], rv) <= TOL     # momentum 0.1, unbiased variance (torch updated rm / rv in place)


@pytest.mark.parametrize("B,H,C", [(3, 112, 64), (2, 31, 64), (4, 16, 128)])
def test_maxpool_forward_backward_kernels(dev, B, H, C):
    from deephisto_amd._lib import check, lib
    g = torch.Generator().manual_seed(H)
    x = torch.randn(B, C, H, H, generator=g)
    x[:, ::2] = (x[:, ::2] * 2).round() / 2     # half of the channels quantised: plenty of exact ties inside the 3x3 windows
    x = x.double().requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (want,) = torch.autograd.grad(y, x, dy)
    xd, dyd = _nhwc(x.detach(), dev), _nhwc(dy, dev)
    Ho = y.shape[2]
    yd = torch.empty(B, Ho, Ho, C, dtype=torch.float32, device=dev)
    dxd = torch.empty_like(xd)
    check(lib().dh_debug_maxpool_f32(xd.data_ptr(), yd.data_ptr(), dyd.data_ptr(), dxd.data_ptr(), B, H, H, C, None), "maxpool")
    assert torch.equal(_nchw(yd), y.detach().float())
    assert _rel(_nchw(dxd), want) <= 1e-6      # gradients are routed, not computed: only float32 sums of <= 4 terms


# ---- bf16 engine (train2): the GEMM-shaped kernels on their own ------------------------------------------------------
def _bf(t):
    return t.bfloat16().float()


@pytest.mark.parametrize("M,N,K,stride", [(1000, 64, 64, 1), (3136, 256, 64, 1), (777, 128, 512, 1), (3136, 2048, 512, 1),
                                          (4 * 14 * 14, 512, 256, 2), (2 * 28 * 28, 128, 64, 2)])
def test_gemm1x1_bf16_kernel(dev, M, N, K, stride):
    """out = A . W^T (+ res) on bf16 MFMA with f32 accumulation against float64 on the same bf16 operands: the products are exact,
    only the output rounding to bf16 (2^-9 relative) and the f32 summation order remain."""
    from deephisto_amd._lib import check, lib
    g = torch.Generator().manual_seed(M + N + K)
    w = _bf(torch.randn(N, K, generator=g) * (1.0 / K) ** 0.5)
    res = _bf(torch.randn(M, N, generator=g))
    if stride == 1:
        a = _bf(torch.randn(M, K, generator=g))
        rows, geo = a, (1, 1, 1, 1)
    else:
        B, Ho = (4, 14) if K == 256 else (2, 28)
        full = _bf(torch.randn(B, 2 * Ho, 2 * Ho, K, generator=g))
        a, rows, geo = full, full[:, ::2, ::2, :].reshape(M, K), (Ho, Ho, 2 * Ho, 2 * Ho)
    want = rows.double() @ w.double().T + res.double()
    ad, wd, rd = a.to(dev).bfloat16().contiguous(), w.to(dev).bfloat16().contiguous(), res.to(dev).bfloat16().contiguous()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    check(lib().dh_debug_gemm1x1_bf16(ad.data_ptr(), wd.data_ptr(), rd.data_ptr(), out.data_ptr(), M, N, K, stride, *geo, 1, None), "gemm")
    got = out.float().cpu().double()
    assert float((got - want).abs().max()) <= 2 ** -8 * max(1.0, float(want.abs().max()))
    assert _rel(got, want) <= 3e-3


@pytest.mark.parametrize("ks,stride,cin,cout,B,H", [
    (3, 1, 64, 64, 3, 56), (3, 1, 256, 256, 4, 14), (3, 2, 128, 128, 3, 28), (3, 1, 512, 512, 5, 7), (3, 1, 128, 128, 2, 12),
    (1, 1, 64, 256, 3, 56), (1, 1, 256, 64, 2, 28),          # 64 x 64 tiles (64-channel layers)
    (1, 1, 512, 128, 3, 28), (1, 1, 256, 1024, 4, 14), (1, 1, 2048, 512, 5, 7), (1, 2, 256, 512, 3, 28), (1, 1, 1024, 256, 2, 6),   # 128 x 128 tiles
])
def test_wgrad_bf16_kernels(dev, ks, stride, cin, cout, B, H):
    """dW through the transposed-LDS-read kernels (3x3 tap groups, 1x1 with 64 x 64 and 128 x 128 workgroup tiles; odd maps, stride 2,
    row pitches that are not multiples of 4) against torch autograd in float64 on the same bf16 operands: relative L2 <= 1e-5
    (exact products, f32 accumulation)."""
    from deephisto_amd._lib import check, lib
    g = torch.Generator().manual_seed(ks * 1000 + cin + cout + H)
    x = _bf(torch.randn(B, cin, H, H, generator=g)).double()
    w = torch.zeros(cout, cin, ks, ks, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, None, stride, ks // 2)
    dz = _bf(torch.randn(y.shape, generator=g)).double()
    (want,) = torch.autograd.grad(y, w, dz)
    xd = x.float().permute(0, 2, 3, 1).contiguous().to(dev).bfloat16()
    dzd = dz.float().permute(0, 2, 3, 1).contiguous().to(dev).bfloat16()
    dw = torch.empty(cout, cin, ks, ks, dtype=torch.float32, device=dev)
    check(lib().dh_debug_wgrad_bf16(dzd.data_ptr(), xd.data_ptr(), dw.data_ptr(), B, H, H, cin, cout, ks, stride, 1, None), "wgrad bf16")
    assert _rel(dw.cpu(), want) <= TOL
    if ks == 3:
        for t in ((0, 0), (0, 2), (2, 0), (2, 2), (1, 1)):
            assert _rel(dw.cpu()[:, :, t[0], t[1]], want[:, :, t[0], t[1]]) <= TOL, t
