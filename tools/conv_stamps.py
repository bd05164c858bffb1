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
DT = sys.argv[2] if len(sys.argv) > 2 else "bf16"
P = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
m = get_model(5, DT).to(dev).eval()
from deephisto_amd import tiles
slide = tiles.synth_slide(4096, 4096, 0, dev)
o = torch.zeros((B, 2), dtype=torch.int32, device=dev)
o[:, 0] = torch.arange(B, device=dev, dtype=torch.int32) % 15 * 256
o[:, 1] = torch.arange(B, device=dev, dtype=torch.int32) // 15 % 15 * 256
m.forward_tiles(slide, o, P); torch.cuda.synchronize()
check(lib().dh_debug_stamps(1, None), "stamps on")
for _ in range(3):
    m.forward_tiles(slide, o, P)
out = np.zeros(64, np.uint64)
check(lib().dh_debug_stamps(0, out.ctypes.data), "stamps read")
names = ["s1 cin64", "s1 cin128", "s1 cin256", "s1 cin512", "s2 cin64", "s2 cin128", "s2 cin256",
         "stem+pool (slots: prefetch, mfma, bn+pool, stores, stage, barrier)"]
print("row           wgs   cyc/wg  cursor%  epil%  mfma%  barr%  prolog% tail%")
for r, n in enumerate(names):
    v = out[8 * r:8 * r + 8].astype(np.float64)
    if v[6] == 0:
        continue
    tot = v[:6].sum()
    ghz = tot / max(v[7], 1) * 0.1   # shader cycles per 100 MHz tick
    clock = f"   clock {ghz:4.2f} GHz" if r < 7 else ""    # the stem kernel records no real-time counter
    print(f"{n:12s} {int(v[6]):5d} {tot / v[6]:8.0f}  " + "  ".join(f"{100 * a / tot:5.1f}" for a in v[:6]) + clock)
