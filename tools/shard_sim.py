#!/usr/bin/env python3
"""What one rank does at world = 1, 2, 4, 8 on the 50 000^2 slide (its tile range + the full ordered accumulate over all the gathered
logits), timed on ONE GPU: an upper bound of the strong-scaling efficiency of predict_full_patched (the RCCL all-gather of
38 416 x 5 floats is not in it).  Tooling only."""
import ctypes as C
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import contextlib
import numpy as np
import torch
from deephisto_amd import tiles
from deephisto_amd._lib import check, lib
from deephisto_amd.examples.predict_full_patched import shard_range
from deephisto_amd.models.patch_cls_simple.model import get_model
from deephisto_amd.patch_samplers.full_samplers import FullImageDenseSampler

dev = torch.device("cuda:0")
side, P = 50000, 256
slide = tiles.synth_slide(side, side, 0, dev)
torch.manual_seed(0)
model = get_model(5, "bf16").to(dev).eval()
with contextlib.redirect_stdout(sys.stderr):
    smp = FullImageDenseSampler(slide, layer=1, patch_size=P, batch_size=64, stride=P, device=dev)
origins, n_unique = smp.origins, smp.n_tiles
h = model.lane_handles(1)[0]
fwd = lib().dh_resnet18_forward_tiles
full = torch.zeros((len(origins), 5), dtype=torch.float32, device=dev)
base = None
for world in (1, 2, 4, 8):
    lo, hi = shard_range(n_unique, world, 0)
    mb = 4096
    if hi - lo > mb:
        mb = -(-(hi - lo) // -(-(hi - lo) // mb))
    def step():
        o_dev = torch.from_numpy(origins[lo:hi]).to(dev)
        local = torch.zeros((hi - lo, 5), dtype=torch.float32, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for s in range(0, hi - lo, mb):
            e = min(s + mb, hi - lo)
            check(fwd(h, slide.data_ptr(), smp.h, smp.w, o_dev.data_ptr() + 8 * s, e - s, P, local.data_ptr() + 20 * s, st), "fwd")
        full[lo:hi] = local                      # stands in for the all-gather
        return tiles.accumulate_logits(full, origins, P, 16, smp.h, smp.w)[1]
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    base = base or dt
    print(f"world {world}: rank time {dt * 1e3:7.2f} ms per slide -> {n_unique / dt / 1e3:7.1f} k patches/s whole job, efficiency <= {base / (world * dt):.3f}", flush=True)
