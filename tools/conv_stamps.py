#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the 3x3 conv kernel (stamped variant). Tooling only."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from deephisto_amd._lib import check, lib
from deephisto_amd.models.patch_cls_simple.model import get_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
m = get_model(5, "bf16").to(dev).eval()
x = torch.rand(B, 3, 256, 256, device=dev)
m(x); torch.cuda.synchronize()
check(lib().dh_debug_stamps(1, None), "stamps on")
for _ in range(3):
    m(x)
out = np.zeros(64, np.uint64)
check(lib().dh_debug_stamps(0, out.ctypes.data), "stamps read")
names = ["s1 cin64", "s1 cin128", "s1 cin256", "s1 cin512", "s2 cin64", "s2 cin128", "s2 cin256"]
print("row           wgs   cyc/wg   load%  mfma%  bar1%  epil%  ldsw%  bar2%")
for r, n in enumerate(names):
    v = out[8 * r:8 * r + 8].astype(np.float64)
    if v[6] == 0:
        continue
    tot = v[:6].sum()
    print(f"{n:12s} {int(v[6]):5d} {tot / v[6]:8.0f}  " + "  ".join(f"{100 * a / tot:5.1f}" for a in v[:6]))
