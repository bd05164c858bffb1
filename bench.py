#!/usr/bin/env python3
"""bench.py -- headline benchmark: 256x256 patches/sec of predict_full_patched on one
synthetic 50000x50000x3 whole-slide image (BASELINE.json configs[2]; sharded over N
GPUs = configs[3]).

One "step" = one whole-slide prediction: every tile of the reference-ordered dense grid
(38 416 tiles at patch 256 / stride 256) goes through fused gather + ResNet-18 forward
(HIP, bf16 MFMA) in micro-batches, logits are all-gathered over RCCL when N > 1, and the
ordered accumulate + argmax produce the int64 class map.  The slide is generated in HBM
(closed form) before the timed region; nothing is read from the host inside it.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task description) with two extra objects:
`roofline` for the dominant kernel (3x3 stride-1 conv, MFMA-bound) timed live with HIP
events on the launch stream, and `cpu_baseline`: the oracle's CPU restatement of the
reference path (NumPy tiling + torch-CPU ResNet-18 fp32) timed on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}  # dense, /opt/skills/guides/MI355X_MICROARCH.md
FLOP_PER_TILE_256 = 4.7375e9                        # SURVEY.md section 8d (2 x 2.3687 GMAC)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--slide", type=int, default=50000, help="slide side (pixels)")
    ap.add_argument("--patch", type=int, default=256)
    ap.add_argument("--stride", type=int, default=256)
    ap.add_argument("--batch", type=int, default=64, help="sampler batch size (grid padding unit)")
    ap.add_argument("--micro-batch", type=int, default=1024, help="tiles per kernel launch")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams (micro-batches in flight)")
    ap.add_argument("--downscale", type=int, default=16)
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=10,
                    help="extra leg: time this many fused training steps (BASELINE configs[1]); 0 = skip")
    return ap.parse_args()


def host_cores() -> int:
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args, budget_s: float) -> dict:
    """The oracle (port of the reference's CPU path a1->a8) on a bounded sample: the first
    tiles of an 8192x8192 closed-form slide, whole batches, until the budget is spent."""
    from oracle import resnet18 as oracle_net
    from oracle import synth, tiling

    threads = host_cores()
    torch.set_num_threads(threads)
    side = 8192
    host = synth.synth_slide(side, side, args.seed)
    net = oracle_net.seeded_model(0, 5).eval()
    batches = tiling.batched_origins(side, side, args.patch, args.stride, args.batch)
    logits, used = [], []
    t0 = time.perf_counter()
    with torch.no_grad():
        for ob in batches:
            x = torch.from_numpy(tiling.features_nchw_predictor(host, ob, args.patch))
            logits.append(net(x).numpy())
            used.append(ob)
            if time.perf_counter() - t0 > budget_s:
                break
    o = np.concatenate(used)
    canvas = tiling.accumulate_logits(side, side, 5, args.downscale, args.patch, o, np.concatenate(logits))
    tiling.class_map(canvas)
    dt = time.perf_counter() - t0
    return {"value": len(o) / dt, "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"first {len(o)} tiles ({len(used)} batches of {args.batch}) of a {side}x{side} "
                      f"closed-form slide, patch {args.patch} stride {args.stride}, torch-CPU fp32 "
                      f"ResNet-18 eager, {threads} threads, {dt:.1f} s"}


def train_leg(dev, steps: int) -> dict:
    """BASELINE configs[1]: models.patch_cls_simple.train step on synthetic annotated regions,
    1 GPU, fp32: batch 64 x 3 x 224 x 224 (config.yaml), HIP forward + CE + backward + Adam."""
    from deephisto_amd import tiles
    from deephisto_amd.models.patch_cls_simple.model import get_model
    from deephisto_amd.patch_samplers.region_samplers import RectRegionRndSampler, synthetic_regions

    side, B, P = 8192, 64, 224
    slide = tiles.synth_slide(side, side, 1, dev)
    smp = RectRegionRndSampler(slide, synthetic_regions(side, side, 5, seed=0), layer=1, patch_size=P, seed=0, device=dev)
    torch.manual_seed(0)
    model = get_model(5, "f32").to(dev).train()
    it = smp.device_batches(B, steps + 2, flips=True)
    for _ in range(2):
        x, y, _c = next(it)
        model.train_step(x, y, lr=1e-4)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for x, y, _c in it:
        loss, _ = model.train_step(x, y, lr=1e-4)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return {"steps_per_s": steps / dt, "samples_per_s": steps * B / dt, "ms_per_step": 1e3 * dt / steps,
            "config": {"workload": "BASELINE configs[1]: train step (fwd + CrossEntropy + bwd + Adam, HIP) on synthetic "
                                   "annotated regions, data assembly (gather + /255 + flips) included",
                       "batch": B, "patch": P, "dtype": "f32", "steps": steps, "last_loss": float(loss)},
            "model_tflops": steps * B * 3 * 3.6271e9 / dt / 1e12}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs a torch.distributed.run launch with {args.gpus} ranks")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    # one rank per GPU; DH_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box) maps every rank to cuda:0 over gloo
    share = os.environ.get("DH_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    from deephisto_amd import tiles
    from deephisto_amd._lib import check, lib
    from deephisto_amd.examples.predict_full_patched import predict_full_patched
    from deephisto_amd.models.patch_cls_simple.model import get_model
    from deephisto_amd.patch_samplers.full_samplers import FullImageDenseSampler

    side = args.slide
    slide = tiles.synth_slide(side, side, args.seed, dev)          # resident in HBM, untimed
    torch.manual_seed(0)
    model = get_model(5, args.dtype).to(dev).eval()                # seeded random init (no checkpoints offline)
    smp = FullImageDenseSampler(slide, layer=1, patch_size=args.patch, batch_size=args.batch,
                                stride=args.stride, device=dev)
    n_tiles = smp.n_tiles

    def step():
        return predict_full_patched(smp, model, 5, downscale=args.downscale, micro_batch=args.micro_batch,
                                    streams=args.streams)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    # live timing of the dominant kernel over the timed region (every 4th launch sampled)
    check(lib().dh_profile_start(4, 65536), "dh_profile_start")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cmap = step()
    fence()
    elapsed = time.perf_counter() - t0
    k_ms, k_flops, k_n = C.c_double(), C.c_double(), C.c_int64()
    check(lib().dh_profile_stop(C.byref(k_ms), C.byref(k_flops), C.byref(k_n)), "dh_profile_stop")
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    assert cmap.shape == (side // args.downscale, side // args.downscale)

    if rank == 0:
        value = args.steps * n_tiles / elapsed
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        achieved = (k_flops.value / (k_ms.value * 1e-3)) / 1e12 if k_ms.value > 0 else 0.0
        flop_tile = FLOP_PER_TILE_256 * (args.patch / 256.0) ** 2
        traffic = None   # HBM bytes per launch of the dominant kernel: from the committed PMC passes (rocprofv3
        pmc = REPO / "profiles" / "r01_pmc_dominant_kernel.json"   # cannot run inside the timed process)
        if pmc.exists() and args.dtype == "bf16" and args.patch == 256:
            doc = json.loads(pmc.read_text())
            if doc.get("micro_batch") == args.micro_batch:
                traffic = doc["traffic_bytes_per_launch"]
        out = {
            "metric": "256x256 patches/sec WSI inference (predict_full_patched)",
            "value": value, "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{2 if world == 1 else 3}]: predict_full_patched on one "
                                   f"{side}x{side}x3 closed-form synthetic WSI resident in HBM, {args.patch}x{args.patch} "
                                   f"tiles stride {args.stride}, ResNet-18 (random init), {args.dtype} MFMA, "
                                   "step = one whole slide (tile grid -> fused gather+forward -> all-gather -> "
                                   "ordered accumulate -> argmax)",
                       "slide_hw": [side, side], "patch": args.patch, "stride": args.stride,
                       "sampler_batch": args.batch, "micro_batch": args.micro_batch, "streams": args.streams, "downscale": args.downscale,
                       "n_tiles": n_tiles, "n_classes": 5,
                       "parallelism": f"tile-range shard x{world}" + (" + RCCL all-gather of logits" if world > 1 else "")},
            "model_tflops": value * flop_tile / 1e12,
            "roofline": {"bound": "mfma", "kernel": f"conv3x3_kernel<{args.dtype}, stride 1, NT=2, 8 waves> (layers 1-3, 10 of 20 convs; two instantiations: layer 1 keeps its weights resident in LDS)",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "launches_timed": int(k_n.value),
                         "avg_launch_us": 1e3 * k_ms.value / max(1, k_n.value),
                         "flops_per_launch": k_flops.value / max(1, k_n.value)},
        }
        if args.train_steps > 0 and world == 1:
            del slide, smp
            torch.cuda.empty_cache()
            out["train"] = train_leg(dev, args.train_steps)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        elif world > 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
