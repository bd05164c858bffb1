#!/usr/bin/env python3
"""A few bf16 forwards of one micro-batch (default 256 tiles) -- workload for PMC passes. Tooling only."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deephisto_amd import tiles
from deephisto_amd.models.patch_cls_simple.model import get_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
m = get_model(5, "bf16").to(dev).eval()
slide = tiles.synth_slide(4096, 4096, 0, dev)
o = torch.zeros((B, 2), dtype=torch.int32, device=dev)
o[:, 0] = torch.arange(B, device=dev, dtype=torch.int32) % 15 * 256
o[:, 1] = torch.arange(B, device=dev, dtype=torch.int32) // 15 % 15 * 256
for _ in range(n):
    m.forward_tiles(slide, o, 256)
torch.cuda.synchronize()
