#!/usr/bin/env python3
"""The stand-alone gather kernels (a4 / a5 path) in GB/s of algorithmic bytes, 1 024 random 256 x 256 tiles of a 50 000^2 slide per launch
(bench.py's `tiler` leg alone). Tooling only."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deephisto_amd import tiles
from deephisto_amd._lib import DH_LAYOUT_NCHW, DH_LAYOUT_NHWC
dev = torch.device("cuda:0")
side, P, n = 50000, 256, 1024
slide = tiles.synth_slide(side, side, 0, dev)
g = torch.Generator().manual_seed(0)
o = torch.stack([torch.randint(0, side - P, (n,), generator=g), torch.randint(0, side - P, (n,), generator=g)], 1).to(torch.int32).to(dev)
for name, layout, dt in (("nhwc_f32", DH_LAYOUT_NHWC, torch.float32), ("nchw_f32", DH_LAYOUT_NCHW, torch.float32), ("nchw_bf16", DH_LAYOUT_NCHW, torch.bfloat16)):
    best = 0.0
    for rep in range(3):
        tiles.gather_tiles(slide, o, P, layout, dt, check_bounds=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            tiles.gather_tiles(slide, o, P, layout, dt, check_bounds=False)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        nbytes = n * (P * P * 3 + P * P * 3 * (4 if dt == torch.float32 else 2))
        best = max(best, nbytes / (ms * 1e-3) / 1e9)
    print(f"{name:10s} {best:7.0f} GB/s  ({best / 8000:.3f} of 8 TB/s)", flush=True)
