#!/usr/bin/env python3
"""f32 conv3x3 through dh_debug_conv_bn_act without residual / ReLU (the dgrad configuration). Tooling only."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
from deephisto_amd._lib import check, lib

dev = torch.device("cuda:0")
for B, hw, cin, cout, relu, use_res in [(8, 7, 512, 512, 0, 0), (4, 8, 512, 512, 0, 0), (3, 14, 512, 256, 0, 0), (8, 5, 512, 512, 0, 0),
                                        (4, 4, 512, 512, 0, 0), (8, 7, 512, 512, 1, 1), (3, 14, 256, 256, 0, 1), (8, 14, 256, 256, 0, 0)]:
    g = torch.Generator().manual_seed(B * hw + cin)
    x = torch.randn(B, cin, hw, hw, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    sc, sh = torch.ones(cout), torch.zeros(cout)
    y = F.conv2d(x, w, None, 1, 1)
    res = torch.randn(y.shape, generator=g)
    want = y + (res if use_res else 0)
    if relu:
        want = F.relu(want)
    x_d = x.permute(0, 2, 3, 1).contiguous().to(dev)
    r_d = res.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.full((B, hw, hw, cout), float("nan"), dtype=torch.float32, device=dev)
    check(lib().dh_debug_conv_bn_act(x_d.data_ptr(), w.contiguous().data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                     r_d.data_ptr() if use_res else None, out.data_ptr(), B, hw, hw, cin, cout, 3, 1, relu, 0, None), "conv")
    got = out.cpu().permute(0, 3, 1, 2)
    d = (got - want).abs()
    print(f"B={B} hw={hw} {cin}->{cout} relu={relu} res={use_res}: max err {float(d.max()):.3e} (scale {float(want.abs().max()):.2f}) "
          f"bad elems {int((d > 1e-3).sum())} nan {int(torch.isnan(got).sum())}", flush=True)
    if int((d > 1e-3).sum()):
        idx = (d > 1e-3).nonzero()
        print("   first bad (b,c,y,x):", idx[:6].tolist(), " imgs:", sorted(set(idx[:, 0].tolist())), " ys:", sorted(set(idx[:, 2].tolist())), " xs:", sorted(set(idx[:, 3].tolist())))
