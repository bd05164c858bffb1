#!/usr/bin/env python3
"""Run a few fused training steps (for rocprofv3 --kernel-trace). Tooling only."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deephisto_amd.models.patch_cls_simple.model import get_model
B, P = 64, 224
ARCH = sys.argv[1] if len(sys.argv) > 1 else "resnet18"      # resnet18 (f32 engine) | resnet50 / resnet18bf16 (bf16 engine)
dev = torch.device("cuda:0")
m = (get_model(5, "f32") if ARCH == "resnet18" else get_model(5, "bf16", arch=ARCH.replace("bf16", ""))).to(dev).train()
x = torch.rand(B, 3, P, P, device=dev)
y = torch.randint(0, 5, (B,), device=dev)
for _ in range(6):
    m.train_step(x, y)
torch.cuda.synchronize()
