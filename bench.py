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

Rank 0 prints ONE JSON line (contract in the task description) with extra objects:
`roofline` for the dominant kernel (3x3 stride-1 conv on 512-pixel tiles: layers 1-4, MFMA-bound) timed live with HIP
events on the launch stream; `cpu_baseline`: the oracle's CPU restatement of the reference
path (NumPy tiling + torch-CPU ResNet-18 fp32) timed on a bounded sample; and, at N = 1,
  `f32`        the same slide in float32 (the mode of north_star's "logits within 1e-4"), its own roofline vs 157.3 TF;
  `train`      BASELINE configs[1]: fused ResNet-18 f32 training steps/s at 64 x 224^2, with a roofline object;
  `train_bf16` the same network trained on the bf16 engine (bf16 MFMA, f32 master weights): informational, configs[1] is fp32;
  `train_r50`  BASELINE configs[4] per-rank work: ResNet-50 bf16 training steps/s at 64 x 224^2 vs the bf16 MFMA peak;
  `tiler`      the stand-alone gather kernels (a4 path) in GB/s against the 8 TB/s HBM peak;
  `cpu_baselines` sampler-only `generator_torch` (one process, as INMEMORY_SINGLEPROC) and a CPU train step, each with cores.
  `p224`       the reference's OWN geometry (examples/predict_full_patched.py:157-167, config.yaml:23): the same slide at 224 x 224 tiles,
               stride 112 (198 916 tiles), patches/s, whole-model fraction at 3.6271 GFLOP per tile, dominant-kernel roofline.
At N > 1 the line also carries `ranks` (what RCCL saw: rank, device index, PCI bus id / uuid and tile range of every rank, gathered over
the group), `backend`, `nccl_version`, `allgather_ms`, and `train_ddp`: ResNet-50 bf16 data-parallel steps/s with the bucketed RCCL all-reduce
(configs[4]) overlapped with the backward pass AND (`overlap_off`) exchanged after it -- the difference shows whether RCCL's kernels got CUs beside
the persistent convolutions.  A failure in that leg is reported in the object; the ranks agree on it through the rendezvous store (never through a
collective a failed rank may no longer take part in) before anyone enters the closing barrier.
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
    ap.add_argument("--micro-batch", type=int, default=4096,
                    help="upper bound of the tiles per kernel launch (the library's maximum; 38 416 tiles run as 9 launches of 3 968 + one of 2 704 -- multiples of 128: measured "
                         "190.8 k patches/s at 1024, 194.7 k at 2048, 196.6 k at 4096 on one box -- prologue and tail of the persistent "
                         "kernels amortised over more tiles)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams (micro-batches in flight)")
    ap.add_argument("--downscale", type=int, default=16)
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=60,
                    help="extra legs: time this many fused training steps (configs[1] f32 ResNet-18, configs[4] bf16 ResNet-50); 0 = skip")
    ap.add_argument("--f32-steps", type=int, default=2, help="extra leg: whole slides in float32 (0 = skip)")
    ap.add_argument("--no-extra-legs", action="store_true", help="headline measurement only (profiling runs)")
    ap.add_argument("--p224-steps", type=int, default=2, help="extra leg: whole slides at the reference's 224 / 112 geometry (0 = skip)")
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


ADAM_BYTES_R18 = 7 * 11_179_077 * 4   # read p, g, m, v + write p, m, v (~313 MB per step)
HBM_PEAK_GBS = 8000.0                 # /opt/skills/guides/MI355X_MICROARCH.md (6 290 GB/s measured for a float4 copy)


def train_hbm_bytes(arch: str, dtype: str, B: int, P: int) -> dict:
    """Algorithmic HBM bytes of one training step as the engines structure it (every tensor counted once per kernel
    that reads or writes it; operand re-reads inside a kernel, weights and the slab buffers of the weight gradients are
    left out).  Per conv + BN: forward  conv (in, Z) | statistics (Z; none for a bf16 1x1 conv: GEMM epilogue) | apply
    (Z [, identity], Y);  backward  BN reduce (Z, dY [, Y]) | BN apply (Z, dY [, Y], dZ [, masked copy]) | wgrad (X, dZ)
    | dgrad (dZ, dX [, identity gradient]).  The stem (round 3): conv (x, Z) | statistics (Z) | BN + ReLU + max-pool in one pass
    (Z, pooled, positions); backward BN reduce and BN apply gather the pooled gradient themselves (Z, pooled gradient, positions
    [, dZ]) | wgrad (x, dZ).  Plus Adam (7 x 4 B per parameter) and the weight re-packing."""
    e = 4 if dtype == "f32" else 2
    H1 = (P + 6 - 7) // 2 + 1
    H2 = (H1 + 2 - 3) // 2 + 1
    convs = [(3, 64, 7, P, H1, False, False)]          # (cin, cout, ks, Hi, Ho, joins an identity, has downsample input)
    h, cin = H2, 64
    if arch == "resnet18":
        for s, c in enumerate((64, 128, 256, 512)):
            for blk in range(2):
                st = 2 if (blk == 0 and s > 0) else 1
                ho = (h + 2 - 3) // st + 1
                convs += [(cin, c, 3, h, ho, False, False), (c, c, 3, ho, ho, True, False)]
                if st != 1 or cin != c:
                    convs.append((cin, c, 1, h, ho, False, True))
                h, cin = ho, c
        n_params = 11_179_077
    else:
        for s, (w, nb) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3))):
            for blk in range(nb):
                st = 2 if (blk == 0 and s > 0) else 1
                ho = (h + 2 - 3) // st + 1
                # bf16 engine: the 3x3 conv's BN gets its backward sums from the dgrad GEMM of conv3 ("fused"); the join BN of a block
                # whose successor has no downsample gets them from that successor's first dgrad GEMM, which also applies the ReLU
                # mask ("premasked": no reduce pass, the apply pass reads neither Y nor writes the masked copy)
                convs += [(cin, w, 1, h, h, False, False), (w, w, 3, h, ho, False, "fused"),
                          (w, 4 * w, 1, ho, ho, True, "premasked" if blk + 1 < nb else False)]
                if st != 1 or cin != 4 * w:
                    convs.append((cin, 4 * w, 1, h, ho, False, True))
                h, cin = ho, 4 * w
        n_params = 23_518_277
    act = 0
    for ci, co, ks, hi, ho, join, ds in convs:
        I, O = B * hi * hi * ci * (4 if ks == 7 else e), B * ho * ho * co * e
        if ks == 7:
            Op, Ib = B * H2 * H2 * 64 * e, B * H2 * H2 * 64
            act += (I + O) + O + (O + Op + Ib) + (O + Op + Ib) + (O + Op + Ib + O) + (I + O)
            continue
        fused_stats = dtype == "bf16" and ks == 1
        fwd = I + O + (0 if fused_stats else O) + O + (O if join else 0) + O
        if ds is True and dtype == "bf16":   # the downsample branch's BN is applied inside the join BN's pass: no apply pass of its own
            fwd -= 2 * O
        if ds == "premasked":
            bwd = 0 + 2 * O + O + (I + O) + (O + I)
        else:
            reduce_pass = 0 if ds == "fused" else (2 + (1 if join else 0)) * O
            bwd = reduce_pass + (2 + (1 if join else 0)) * O + O + (O if join else 0) + (I + O) + (O + I)
        act += fwd + bwd
    adam = 7 * 4 * n_params
    pack = (4 + 2 * e) * n_params
    return {"activation_passes": act, "adam": adam, "weight_packing": pack, "total": act + adam + pack}



def train_leg(dev, steps: int, arch: str = "resnet18", dtype: str = "f32", group=None) -> dict:
    """Fused training steps (forward + CrossEntropy + backward [+ bucketed all-reduce] + Adam, all HIP) on synthetic
    annotated regions, batch 64 x 3 x 224 x 224 per rank (config.yaml), data assembly (gather + /255 + flips) included.
    resnet18 / f32 = BASELINE configs[1]; resnet50 / bf16 = configs[4] (per-rank work; `group` set => data parallel)."""
    from deephisto_amd import tiles
    from deephisto_amd.models.patch_cls_simple.model import get_model
    from deephisto_amd.patch_samplers.region_samplers import RectRegionRndSampler, synthetic_regions

    side, B, P = 8192, 64, 224
    rank = int(os.environ.get("RANK", "0"))
    slide = tiles.synth_slide(side, side, 1, dev)
    smp = RectRegionRndSampler(slide, synthetic_regions(side, side, 5, seed=0), layer=1, patch_size=P, seed=rank, device=dev)
    torch.manual_seed(0)
    model = get_model(5, dtype, arch=arch).to(dev).train()
    it = smp.device_batches(B, steps + 2, flips=True)
    for _ in range(2):
        x, y, _c = next(it)
        model.train_step(x, y, lr=1e-4, group=group)
    torch.cuda.synchronize(dev)
    if group is not None:
        import torch.distributed as dist
        dist.barrier()
    t0 = time.perf_counter()
    for x, y, _c in it:
        loss, _ = model.train_step(x, y, lr=1e-4, group=group)
    torch.cuda.synchronize(dev)
    if group is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    world = 1 if group is None else dist.get_world_size(group)
    fwd_flop = (3.6271e9 if arch == "resnet18" else 8.1743e9)    # per 224^2 sample
    tflops = world * steps * B * 3 * fwd_flop / dt / 1e12        # train step counted as 3 x forward
    peak = MFMA_PEAK_TFLOPS[dtype] * world
    out = {"steps_per_s": steps / dt, "samples_per_s": world * steps * B / dt, "ms_per_step": 1e3 * dt / steps,
           "config": {"workload": f"train step (fwd + CrossEntropy + bwd + Adam, HIP) of {arch} in {dtype} on synthetic annotated "
                                  "regions, data assembly (gather + /255 + flips) included",
                      "arch": arch, "batch_per_rank": B, "patch": P, "dtype": dtype, "steps": steps, "ranks": world,
                      "last_loss": float(loss)},
           "model_tflops": tflops,
           "roofline": {"bound": "mfma", "achieved": tflops, "peak": peak, "unit": "TFLOP/s", "frac": tflops / peak,
                        "flop_per_sample": 3 * fwd_flop,
                        "note": "whole step (3 x forward FLOPs) over wall time, data assembly and optimizer included"}}
    hb = train_hbm_bytes(arch, dtype, B, P)
    gbs = world * hb["total"] * steps / dt / 1e9
    out["roofline"].update({"hbm_bytes_per_step": hb["total"], "hbm_bytes_breakdown": hb, "hbm_achieved_gbs": gbs,
                            "hbm_peak_gbs": HBM_PEAK_GBS * world, "hbm_frac": gbs / (HBM_PEAK_GBS * world),
                            "hbm_note": "algorithmic bytes of the engine's passes (train_hbm_bytes) over wall time: the BN / pooling / "
                                        "Adam passes are HBM-bound, the convolutions MFMA-bound; both fractions describe the same step"})
    if world > 1:
        from deephisto_amd.models.patch_cls_simple.ddp import default_wire
        out["config"]["parallelism"] = f"dp{world}: bucketed (~25 MB) {default_wire()} all-reduce overlapped with backward (DH_DDP_WIRE=bf16 halves the bytes)"
        out["buckets"] = [c for _, _, c in getattr(getattr(model, "_engine", model), "overlap_log", [])]
    return out


def tiler_leg(dev, slide, args) -> dict:
    """The stand-alone gather kernels (a4: generator_torch features NHWC f32; a5: batch_predictor input NCHW f32 / bf16):
    algorithmic bytes per 256^2 tile = 196 608 read + 786 432 (f32) / 393 216 (bf16) written."""
    from deephisto_amd import tiles
    from deephisto_amd._lib import DH_LAYOUT_NCHW, DH_LAYOUT_NHWC

    P, n = args.patch, 1024
    g = torch.Generator().manual_seed(0)
    o = torch.stack([torch.randint(0, slide.shape[0] - P, (n,), generator=g), torch.randint(0, slide.shape[1] - P, (n,), generator=g)], 1)
    o = o.to(torch.int32).to(dev)
    out = {}
    for name, layout, dt in (("nhwc_f32", DH_LAYOUT_NHWC, torch.float32), ("nchw_f32", DH_LAYOUT_NCHW, torch.float32),
                             ("nchw_bf16", DH_LAYOUT_NCHW, torch.bfloat16)):
        tiles.gather_tiles(slide, o, P, layout, dt, check_bounds=False)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            tiles.gather_tiles(slide, o, P, layout, dt, check_bounds=False)
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / 5
        nbytes = n * (P * P * 3 + P * P * 3 * (4 if dt == torch.float32 else 2))
        out[name] = {"tiles_per_s": n / (ms * 1e-3), "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": nbytes / (ms * 1e-3) / 1e9 / 8000.0, "bytes_per_tile": nbytes // n}
    return {"bound": "hbm", "tiles_per_launch": n, "patch": P, "kernels": out}


def cpu_sampler_baseline(args, budget_s: float) -> dict:
    """the oracle's restatement of FullImageDenseSampler.generator_torch
    (full_samplers.py:437-452: stack -> astype(f32)/255 -> torch.tensor, coords) in ONE process, as INMEMORY_SINGLEPROC."""
    from oracle import synth, tiling

    torch.set_num_threads(1)
    side = 4096
    host = synth.synth_slide(side, side, args.seed)
    batches = tiling.batched_origins(side, side, args.patch, args.stride, args.batch)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:          # whole passes over the slide until the budget is spent
        for ob in batches:
            f = torch.tensor(tiling.features_nhwc(host, ob, args.patch))
            c = torch.tensor(ob.astype(np.float32))
            n += len(ob)
    dt = time.perf_counter() - t0
    del f, c
    return {"value": n / dt, "unit": "patches/s", "cores": 1, "kind": "port",
            "sample": f"{n} tiles ({n // 256} passes over a {side}x{side} closed-form slide), patch {args.patch}, batch {args.batch}, one process, {dt:.1f} s"}


def cpu_train_baseline(steps: int, arch: str = "resnet18") -> dict:
    """torch-CPU eager ResNet-18 / ResNet-50 restatement, CrossEntropyLoss, Adam(lr=1e-4), all cores."""
    from oracle import resnet18 as o18
    from oracle import resnet50 as o50

    oracle_net = o18 if arch == "resnet18" else o50
    threads = host_cores()
    torch.set_num_threads(threads)
    net = oracle_net.seeded_model(0, 5).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(64, 3, 224, 224, generator=g)
    y = torch.randint(0, 5, (64,), generator=g)
    o18.train_step(net, opt, x, y)
    t0 = time.perf_counter()
    for _ in range(steps):
        o18.train_step(net, opt, x, y)
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": f"{steps} steps of batch 64 x 224^2, torch-CPU fp32 {arch} eager + Adam, {threads} threads, {dt:.1f} s"}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher's environment: start the N ranks as a CHILD `torch.distributed.run` job
    (one process per GPU, rendezvous on 127.0.0.1) before this process has touched a GPU, relay rank 0's JSON line on stdout
    and return the job's exit code.  (Never an exec: a process that may already have initialised the GPU must not be replaced.)"""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:            # ranks other than 0 print nothing on stdout; keep the last JSON object line
        if out.lstrip().startswith("{"):
            line = out
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    return rc if rc != 0 or line is not None else 1


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit(self_launch(args.gpus))   # `python bench.py --gpus N`: become the launcher, before any GPU call
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
        from deephisto_amd.distributed import dist_timeout
        # explicit collective timeout (DH_DIST_TIMEOUT_S, default 600 s): a stuck peer ends the job instead of holding it for the backend's default
        if share:
            dist.init_process_group("gloo", timeout=dist_timeout())
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=dist_timeout())  # RCCL over xGMI

    from deephisto_amd import tiles
    from deephisto_amd._lib import check, lib
    from deephisto_amd.examples.predict_full_patched import predict_full_patched
    from deephisto_amd.models.patch_cls_simple.model import get_model
    from deephisto_amd.patch_samplers.full_samplers import FullImageDenseSampler

    side = args.slide
    slide = tiles.synth_slide(side, side, args.seed, dev)          # resident in HBM, untimed
    torch.manual_seed(0)
    model = get_model(5, args.dtype).to(dev).eval()                # seeded random init (no checkpoints offline)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):   # the sampler prints the slide size like the reference's constructor; stdout carries ONE JSON line
        smp = FullImageDenseSampler(slide, layer=1, patch_size=args.patch, batch_size=args.batch,
                                    stride=args.stride, device=dev)
    n_tiles = smp.n_tiles

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    exchange_events: list = []   # (start, end) HIP events around every all-gather of the timed region (world > 1)

    def timed_predict(mdl, steps, warmup, micro_batch=None, sampler=None):
        """(elapsed seconds for `steps` whole slides, dominant-kernel ms / flops / samples over the timed region, class map)"""
        def step():
            return predict_full_patched(sampler or smp, mdl, 5, downscale=args.downscale, micro_batch=micro_batch or args.micro_batch, streams=args.streams,
                                        timing=exchange_events)
        for _ in range(warmup):
            step()
        fence()
        # live timing of the dominant kernel over the timed region (every 4th launch sampled)
        check(lib().dh_profile_start(4, 65536), "dh_profile_start")
        exchange_events.clear()
        t0 = time.perf_counter()
        for _ in range(steps):
            cm = step()
        fence()
        el = time.perf_counter() - t0
        k_ms, k_flops, k_n = C.c_double(), C.c_double(), C.c_int64()
        check(lib().dh_profile_stop(C.byref(k_ms), C.byref(k_flops), C.byref(k_n)), "dh_profile_stop")
        return el, k_ms.value, k_flops.value, int(k_n.value), cm

    elapsed, k_ms_v, k_flops_v, k_n_v, cmap = timed_predict(model, args.steps, args.warmup)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    assert cmap.shape == (side // args.downscale, side // args.downscale)

    # what RCCL saw (N > 1): every rank's device and tile range, gathered over the group itself
    ranks_seen = None
    if world > 1:
        from deephisto_amd.examples.predict_full_patched import shard_range
        props = torch.cuda.get_device_properties(dev_index)
        lo, hi = shard_range(n_tiles, world, rank)
        mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "name": props.name,
                "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")) or None,
                "tile_lo": lo, "tile_hi": hi, "pid": os.getpid()}
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, mine)

    # configs[4] under torch.distributed: ResNet-50 bf16 data-parallel steps (every rank takes part; reported by rank 0)
    allgather_ms = None
    if world > 1 and exchange_events:
        allgather_ms = sum(a.elapsed_time(b) for a, b in exchange_events) / len(exchange_events)
    train_ddp = None
    if world > 1 and args.train_steps > 0 and not args.no_extra_legs:
        try:
            del slide, smp
            torch.cuda.empty_cache()
            if os.environ.get("DH_BENCH_FAIL_DDP") == "1":   # test hook: the leg fails on every rank before its first collective
                raise RuntimeError("DH_BENCH_FAIL_DDP=1: injected failure of the train_ddp leg")
            # the same per-rank step three ways: gradients exchanged in float32 (default wire), in bf16 (half the bytes per xGMI
            # link), and not exchanged at all -- the difference is what the bucketed, overlapped all-reduce leaves exposed
            os.environ["DH_DDP_WIRE"] = "f32"
            train_ddp = train_leg(dev, args.train_steps, "resnet50", "bf16", group=dist.group.WORLD)
            os.environ["DH_DDP_WIRE"] = "bf16"
            leg_bf16 = train_leg(dev, args.train_steps, "resnet50", "bf16", group=dist.group.WORLD)
            os.environ["DH_DDP_WIRE"] = "f32"
            # the same buckets exchanged AFTER the backward pass (DH_DDP_OVERLAP=0; bit-equal gradients): if this is not slower than the
            # overlapped step, RCCL's kernels did not get CUs beside the persistent convolution kernels and the overlap hides nothing
            os.environ["DH_DDP_OVERLAP"] = "0"
            leg_late = train_leg(dev, args.train_steps, "resnet50", "bf16", group=dist.group.WORLD)
            os.environ["DH_DDP_OVERLAP"] = "1"
            solo = [dist.new_group([r]) for r in range(world)][rank]   # a group of this rank alone: train_step sees world = 1, no exchange
            local = train_leg(dev, args.train_steps, "resnet50", "bf16", group=solo)
            lm = torch.tensor([local["ms_per_step"]], dtype=torch.float64, device=dev)
            dist.all_reduce(lm, op=dist.ReduceOp.MAX)     # the slowest rank's un-exchanged step
            train_ddp["local_step_ms"] = float(lm.item())
            train_ddp["allreduce_exposed_ms"] = train_ddp["ms_per_step"] - float(lm.item())
            train_ddp["overlap_off"] = {"steps_per_s": leg_late["steps_per_s"], "ms_per_step": leg_late["ms_per_step"],
                                        "allreduce_exposed_ms": leg_late["ms_per_step"] - float(lm.item()),
                                        "note": "DH_DDP_OVERLAP=0: same buckets, one after the other, after the backward pass"}
            train_ddp["wire_bf16"] = {"steps_per_s": leg_bf16["steps_per_s"], "ms_per_step": leg_bf16["ms_per_step"],
                                      "allreduce_exposed_ms": leg_bf16["ms_per_step"] - float(lm.item()),
                                      "parallelism": leg_bf16["config"].get("parallelism")}
        except Exception as e:   # never costs the headline number
            err = {"error": f"{type(e).__name__}: {e}"[:400]}
            train_ddp = dict(train_ddp, **err) if isinstance(train_ddp, dict) else err
    stuck = None
    if world > 1:
        # Agree on how the leg ended BEFORE any further collective: a rank that failed inside the leg has left collectives
        # its peers may still be waiting in, so a barrier here could mismatch and block until the backend timeout.  The agreement goes
        # through the rendezvous store (host side, no GPU, bounded wait): when every rank has checked in, nobody is inside a collective
        # and the barrier is safe; when some rank does not arrive, rank 0 still prints its line and everybody leaves non-zero.
        store = dist.distributed_c10d._get_default_store()
        my_err = train_ddp.get("error") if isinstance(train_ddp, dict) else None
        store.set(f"bench/leg_done/{rank}", my_err or "ok")
        from datetime import timedelta
        try:
            store.wait([f"bench/leg_done/{r}" for r in range(world)], timedelta(seconds=float(os.environ.get("DH_BENCH_AGREE_S", "120"))))
            states = [store.get(f"bench/leg_done/{r}").decode() for r in range(world)]
            bad = {r: st for r, st in enumerate(states) if st != "ok"}
            if bad and isinstance(train_ddp, dict) and "error" not in train_ddp:
                train_ddp["error"] = f"rank(s) {sorted(bad)} failed: {next(iter(bad.values()))}"[:400]
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:   # some rank never left the leg
            stuck = f"{type(e).__name__}: not every rank finished the train_ddp leg: {e}"[:300]
            train_ddp = dict(train_ddp or {}, error=stuck)

    if rank == 0:
        value = args.steps * n_tiles / elapsed
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        achieved = (k_flops_v / (k_ms_v * 1e-3)) / 1e12 if k_ms_v > 0 else 0.0
        flop_tile = FLOP_PER_TILE_256 * (args.patch / 256.0) ** 2
        traffic = None   # HBM bytes per launch of the dominant kernel: from the committed PMC passes (rocprofv3
        pmc = next((q for q in (REPO / "profiles" / f"r0{r}_pmc_dominant_kernel.json" for r in (5, 4, 3, 2, 1)) if q.exists()), REPO / "none")   # cannot run inside the timed process)
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
            "roofline": {"bound": "mfma", "kernel": f"conv3x3_kernel<{args.dtype}, stride 1, NT=2, 8 waves> (every stride-1 3x3 conv of layers 1-4: 13 of 20 convs, 84 % of the FLOPs; two instantiations: layer 1 keeps its weights resident in LDS)",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "traffic_source": (f"profiles/{pmc.name} (two separate rocprofv3 --pmc passes of this command, FETCH_SIZE x 2 + "
                                            "WRITE_SIZE, per launch; not measured in this run)") if traffic is not None else None,
                         "launches_timed": k_n_v,
                         "avg_launch_us": 1e3 * k_ms_v / max(1, k_n_v),
                         "flops_per_launch": k_flops_v / max(1, k_n_v)},
        }
        extra = world == 1 and not args.no_extra_legs
        if extra and args.f32_steps > 0 and args.dtype == "bf16":
            # the configuration north_star's "logits within 1e-4 of the CPU reference" belongs to: same slide, float32 MFMA
            m32 = get_model(5, "f32").to(dev).eval()
            m32.load_state_dict(model.state_dict())
            el, kms, kfl, kn, cm32 = timed_predict(m32, args.f32_steps, 1, micro_batch=min(args.micro_batch, 1024))   # float32 activations: 4 GiB per tensor at 4096 tiles
            a32 = (kfl / (kms * 1e-3)) / 1e12 if kms > 0 else 0.0
            v32 = args.f32_steps * n_tiles / el
            out["f32"] = {"value": v32, "unit": "patches/s", "steps": args.f32_steps, "ms_per_step": 1e3 * el / args.f32_steps,
                          "dtype": "f32", "model_tflops": v32 * flop_tile / 1e12,
                          "class_map_agreement_with_bf16": float((cm32 == cmap).float().mean()),
                          "roofline": {"bound": "mfma", "kernel": "conv3x3_kernel<f32, stride 1, NT=2, 8 waves> (v_mfma_f32_32x32x2_f32)",
                                       "achieved": a32, "peak": MFMA_PEAK_TFLOPS["f32"], "unit": "TFLOP/s",
                                       "frac": a32 / MFMA_PEAK_TFLOPS["f32"], "launches_timed": kn,
                                       "avg_launch_us": 1e3 * kms / max(1, kn)}}
            del m32
        if extra and args.p224_steps > 0 and args.dtype == "bf16" and (args.patch, args.stride) != (224, 112):
            # the reference's own call sites tile 224 x 224 at stride 112 (examples/predict_full_patched.py:157-167, config.yaml:23)
            with contextlib.redirect_stdout(sys.stderr):
                smp224 = FullImageDenseSampler(slide, layer=1, patch_size=224, batch_size=args.batch, stride=112, device=dev)
            el, kms, kfl, kn, cm224 = timed_predict(model, args.p224_steps, 1, sampler=smp224)
            assert cm224.shape == (side // args.downscale, side // args.downscale)
            v224 = args.p224_steps * smp224.n_tiles / el
            f224 = 3.6271e9                                  # FLOP per 224^2 tile
            a224 = (kfl / (kms * 1e-3)) / 1e12 if kms > 0 else 0.0
            out["p224"] = {"value": v224, "unit": "patches/s", "steps": args.p224_steps, "ms_per_step": 1e3 * el / args.p224_steps,
                           "config": {"workload": f"predict_full_patched on the same {side}x{side} slide at the reference's geometry: 224x224 tiles, "
                                                  "stride 112 (examples/predict_full_patched.py:157-167)", "patch": 224, "stride": 112,
                                      "n_tiles": smp224.n_tiles, "micro_batch": args.micro_batch},
                           "model_tflops": v224 * f224 / 1e12, "model_frac": v224 * f224 / 1e12 / peak, "flop_per_tile": f224,
                           "roofline": {"bound": "mfma", "kernel": "conv3x3_kernel<bf16, stride 1> at 56 / 28 / 14-pixel maps (the launches the library samples as dominant)",
                                        "achieved": a224, "peak": peak, "unit": "TFLOP/s", "frac": a224 / peak, "launches_timed": kn,
                                        "avg_launch_us": 1e3 * kms / max(1, kn), "flops_per_launch": kfl / max(1, kn)}}
            del smp224
        if extra:
            out["tiler"] = tiler_leg(dev, slide, args)
        if extra and args.train_steps > 0:
            del slide, smp
            torch.cuda.empty_cache()
            out["train"] = train_leg(dev, args.train_steps, "resnet18", "f32")
            out["train_bf16"] = train_leg(dev, args.train_steps, "resnet18", "bf16")   # the same network on the bf16 engine (f32 masters)
            out["train_r50"] = train_leg(dev, args.train_steps, "resnet50", "bf16")
        if world > 1:
            out["ranks"] = ranks_seen            # gathered over the group: what RCCL saw (device per rank, tile range per rank)
            out["backend"] = "gloo (DH_BENCH_SHARE_GPU rehearsal: every rank on cuda:0)" if share else "nccl (RCCL)"
            try:
                out["nccl_version"] = ".".join(map(str, torch.cuda.nccl.version()))
            except Exception:
                out["nccl_version"] = None
            out["distinct_devices"] = len({(r or {}).get("pci_bus_id") or (r or {}).get("uuid") or (r or {}).get("device") for r in ranks_seen})
            out["allgather_ms"] = allgather_ms   # HIP-event time of the ONE RCCL all-gather of per-tile logits, mean per slide (rank 0)
            out["train_ddp"] = train_ddp
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)   # rank 0, after the group is gone
            if not args.no_extra_legs and world == 1:
                out["cpu_baselines"] = {"sampler_generator_torch": cpu_sampler_baseline(args, 4.0),
                                        "train_step_resnet18_f32": cpu_train_baseline(2),
                                        "train_step_resnet50_f32": cpu_train_baseline(2, "resnet50")}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if stuck is not None:   # a rank is still inside a collective: leave without one, non-zero, so that the launcher ends the job now
        sys.stderr.write(f"[bench rank {rank}] {stuck}\n")
        sys.stderr.flush()
        os._exit(3)


if __name__ == "__main__":
    main()
