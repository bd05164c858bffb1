#!/usr/bin/env python3
"""How much of a training step is batch assembly?  fixed batch vs the sampler's device_batches. Tooling only."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deephisto_amd import tiles
from deephisto_amd.models.patch_cls_simple.model import get_model
from deephisto_amd.patch_samplers.region_samplers import RectRegionRndSampler, synthetic_regions
dev = torch.device("cuda:0")
side, B, P, steps = 8192, 64, 224, 60
slide = tiles.synth_slide(side, side, 1, dev)
smp = RectRegionRndSampler(slide, synthetic_regions(side, side, 5, seed=0), layer=1, patch_size=P, seed=0, device=dev)
for arch, dt in (("resnet50", "bf16"), ("resnet18", "bf16"), ("resnet18", "f32")):
    torch.manual_seed(0)
    m = get_model(5, dt, arch=arch).to(dev).train()
    it = smp.device_batches(B, 5, flips=True)
    for x, y, _ in it:
        m.train_step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.train_step(x, y)
    torch.cuda.synchronize()
    fixed = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for x, y, _ in smp.device_batches(B, steps, flips=True):
        m.train_step(x, y)
    torch.cuda.synchronize()
    samp = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    n = 0
    for x, y, _ in smp.device_batches(B, steps, flips=True):
        n += 1
    torch.cuda.synchronize()
    only = (time.perf_counter() - t0) / steps
    print(f"{arch} {dt}: fixed batch {fixed*1e3:.3f} ms/step, with sampler {samp*1e3:.3f} ms/step, sampler alone {only*1e3:.3f} ms/batch", flush=True)
    del m
