#!/usr/bin/env python3
"""Times the bf16 engine's 1x1 GEMM and weight-gradient kernels on the ResNet-50 shapes of a 64 x 224^2 step (debug hooks).
Tooling only."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deephisto_amd._lib import check, lib

dev = torch.device("cuda:0")
REP = 50
print("gemm1x1: M N K  us  TFLOP/s  GB/s(algorithmic in+out)")
for M, N, K in [(200704, 256, 64), (200704, 64, 256), (200704, 64, 64), (50176, 512, 128), (50176, 128, 512), (12544, 1024, 256),
                (12544, 256, 1024), (3136, 2048, 512), (3136, 512, 2048), (50176, 256, 512), (12544, 512, 1024), (3136, 1024, 2048)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    args = (a.data_ptr(), w.data_ptr(), None, out.data_ptr(), M, N, K, 1, 1, 1, 1, 1)
    check(lib().dh_debug_gemm1x1_bf16(*args, 2, None), "gemm")
    t0 = time.perf_counter()
    check(lib().dh_debug_gemm1x1_bf16(*args, REP, None), "gemm")
    us = (time.perf_counter() - t0) / REP * 1e6
    err = float((out[:256].float() - a[:256].float() @ w.float().T).abs().max())
    print(f"{M:7d} {N:5d} {K:5d}  {us:7.1f}  {2 * M * N * K / us / 1e6:7.1f}  {(M * K + M * N) * 2 / us / 1e3:7.0f}   err {err:.3f}")
if "gemm" in sys.argv[1:]:
    sys.exit(0)
print("wgrad: B H cin cout ks  us  TFLOP/s")
for B, H, cin, cout, ks in [(64, 56, 64, 256, 1), (64, 56, 256, 64, 1), (64, 28, 512, 128, 1), (64, 14, 1024, 256, 1), (64, 14, 256, 1024, 1),
                            (64, 7, 2048, 512, 1), (64, 56, 64, 64, 3), (64, 28, 128, 128, 3), (64, 14, 256, 256, 3), (64, 7, 512, 512, 3)]:
    x = torch.randn(B, H, H, cin, device=dev).bfloat16()
    dz = torch.randn(B, H, H, cout, device=dev).bfloat16()
    dw = torch.empty(cout, cin, ks, ks, device=dev)
    args = (dz.data_ptr(), x.data_ptr(), dw.data_ptr(), B, H, H, cin, cout, ks, 1)
    check(lib().dh_debug_wgrad_bf16(*args, 2, None), "wgrad")
    t0 = time.perf_counter()
    check(lib().dh_debug_wgrad_bf16(*args, REP, None), "wgrad")
    us = (time.perf_counter() - t0) / REP * 1e6
    print(f"{B:3d} {H:3d} {cin:5d} {cout:5d} {ks}  {us:7.1f}  {2 * B * H * H * cin * cout * ks * ks / us / 1e6:7.1f}")
