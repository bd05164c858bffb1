#!/usr/bin/env python3
"""Is a fused training step bound by the host's launch rate or by the GPU?  Host time to ENQUEUE a step (no sync) vs wall time per
step with a sync every 10 steps.  Tooling only.  usage: host_bound_check.py [resnet18|resnet50]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deephisto_amd.models.patch_cls_simple.model import get_model
ARCH = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
dev = torch.device("cuda:0")
m = (get_model(5, "f32") if ARCH == "resnet18" else get_model(5, "bf16", arch=ARCH.replace("bf16", ""))).to(dev).train()
x = torch.rand(64, 3, 224, 224, device=dev)
y = torch.randint(0, 5, (64,), device=dev)
for _ in range(3):
    m.train_step(x, y)
torch.cuda.synchronize()
enq = []
t0 = time.perf_counter()
for _ in range(10):
    t1 = time.perf_counter()
    m.train_step(x, y)
    enq.append(time.perf_counter() - t1)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 10
print(f"{ARCH}: host enqueue per step {1e3 * sum(enq) / len(enq):.2f} ms (min {1e3 * min(enq):.2f}), wall per step {1e3 * wall:.2f} ms")
