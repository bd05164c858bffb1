#!/usr/bin/env python3
"""Wall-clock ms per fused training step (64 x 224^2) of the three training configurations. Tooling only.
usage: train_time.py [resnet50 resnet18bf16 resnet18] [--steps N]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deephisto_amd.models.patch_cls_simple.model import get_model
B, P = 64, 224
argv = sys.argv[1:]
steps = 60
if "--steps" in argv:
    i = argv.index("--steps")
    steps = int(argv[i + 1])
    del argv[i:i + 2]
args = [a for a in argv if not a.startswith("--")] or ["resnet50", "resnet18bf16", "resnet18"]
dev = torch.device("cuda:0")
for arch in args:
    torch.manual_seed(0)
    m = (get_model(5, "f32") if arch == "resnet18" else get_model(5, "bf16", arch=arch.replace("bf16", ""))).to(dev).train()
    x = torch.rand(B, 3, P, P, device=dev)
    y = torch.randint(0, 5, (B,), device=dev)
    for _ in range(5):
        loss, _ = m.train_step(x, y)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, _ = m.train_step(x, y)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    print(f"{arch:14s} {best * 1e3:8.3f} ms/step  {1 / best:7.1f} steps/s   loss {float(loss):.4f}", flush=True)
    del m
