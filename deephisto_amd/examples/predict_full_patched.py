"""Whole-slide patched prediction -- drop-in for examples/predict_full_patched.py.

Keeps `ImagePredictorPatched(psim_path, patch_sampler, batch_predictor, anno, layer,
downscale).process()`, `batch_predictor(patches, model, device)` and
`load_model(weights_path, device)` (predict_full_patched.py:22-78, 116-126), and adds
`predict_full_patched(...)`, the device-resident fast path used by bench.py:
tile ranges are sharded over the ranks of a torch.distributed job (RCCL over
xGMI), every rank runs fused gather+ResNet-18 on its range, per-tile logits are
all-gathered, and the ordered accumulation + argmax run once.
"""
from __future__ import annotations

from pathlib import Path
from typing import Callable

import numpy as np
import torch

import ctypes as C
import os

from .. import tiles
from .._lib import DH_LAYOUT_NCHW, check, lib
from ..models.patch_cls_simple.model import ResNet18HIP, get_model
from ..patch_samplers.full_samplers import DevicePatch, FullImageDenseSampler
from ..psimage_compat import Patch, open_slide


def _n_classes(anno) -> int:
    if hasattr(anno, "anno_classes"):  # AnnoDescription (predict_full_patched.py:43)
        return len(anno.anno_classes)
    return int(anno)


class ImagePredictorPatched:
    def __init__(
        self,
        psim_path,
        patch_sampler,
        batch_predictor: Callable[[list[Patch]], "np.ndarray"],
        anno,
        layer: int,
        downscale: int = 4,
        device="cuda",
    ):
        self.patch_sampler = patch_sampler
        self.batch_predictor = batch_predictor
        self.anno = anno
        self.layer = layer
        self.downscale = downscale
        self.device = torch.device(device)
        if isinstance(psim_path, (tuple, list)):  # (h, w) given directly
            self.h, self.w = int(psim_path[0]), int(psim_path[1])
        elif isinstance(psim_path, torch.Tensor):
            self.h, self.w = int(psim_path.shape[0]), int(psim_path.shape[1])
        else:
            with open_slide(psim_path) as psim:
                self.h, self.w = psim.layer_size(self.layer)

    def process(self) -> np.ndarray:
        """int64[h//d, w//d] class map.  Iterates the sampler and the caller's
        batch_predictor like the reference (predict_full_patched.py:47-48); the per-patch
        `prediction[...] += logits` loop and the argmax (:49-62) run on the GPU, in the
        same order (padding duplicates included), once all batches are in."""
        n = _n_classes(self.anno)
        runs: list[tuple[int, list, list]] = []  # (patch_size, origins, logits) per run of equal size
        for patches, _progress in self.patch_sampler:
            preds = self.batch_predictor(patches)
            if isinstance(preds, torch.Tensor):
                preds = preds.detach().to(torch.float32)
            else:
                preds = torch.from_numpy(np.asarray(preds, dtype=np.float32))
            if preds.shape[0] != len(patches) or preds.shape[1] != n:
                raise ValueError(f"batch_predictor returned {tuple(preds.shape)} for {len(patches)} patches, {n} classes")
            for i, p in enumerate(patches):
                if not runs or runs[-1][0] != p.patch_size:
                    runs.append((p.patch_size, [], []))
                runs[-1][1].append((p.pos_y, p.pos_x))
                runs[-1][2].append(preds[i:i + 1])
        canvas, cmap = None, None
        for ps, origins, logit_rows in runs:
            logits = torch.cat(logit_rows).to(self.device).contiguous()
            canvas, cmap = tiles.accumulate_logits(logits, np.asarray(origins, dtype=np.int32), ps,
                                                   self.downscale, self.h, self.w, canvas=canvas)
        if cmap is None:
            return np.zeros((self.h // self.downscale, self.w // self.downscale), dtype=np.int64)
        return cmap.cpu().numpy()


def batch_predictor(patches: list[Patch], model, device) -> np.ndarray:
    """float32[B, n_cls] raw logits for a list of patches (predict_full_patched.py:66-78).

    Patches cut by this package's samplers are gathered from the HBM-resident slide
    inside the stem kernel (no float copy of the pixels exists anywhere); foreign
    patches carrying host arrays are uploaded as uint8 and gathered the same way."""
    device = torch.device(device)
    ps = patches[0].patch_size
    origins = np.array([(p.pos_y, p.pos_x) for p in patches], dtype=np.int32)
    if all(isinstance(p, DevicePatch) and p._sampler is patches[0]._sampler for p in patches):
        slide = patches[0]._sampler.data_device
    else:  # stack foreign host patches into a strip "slide" of B tiles
        slide = torch.from_numpy(np.concatenate([np.asarray(p.data, dtype=np.uint8) for p in patches], axis=0)).to(device)
        origins = np.array([(i * ps, 0) for i in range(len(patches))], dtype=np.int32)
    o_dev = torch.from_numpy(origins).to(slide.device)
    if isinstance(model, ResNet18HIP):
        out = model.forward_tiles(slide, o_dev, ps)
    else:  # any other nn.Module: build the NCHW float input with the gather kernel
        with torch.no_grad():
            out = model(tiles.gather_tiles(slide, o_dev, ps, DH_LAYOUT_NCHW, torch.float32))
    return out.detach().cpu().numpy()


def load_model(weights_path, device, compute_dtype: str = "f32") -> torch.nn.Module:
    """predict_full_patched.py:116-126: 5-class model, state_dict loaded weights_only."""
    model = get_model(n_classes=5, compute_dtype=compute_dtype).to(device)
    model.load_state_dict(torch.load(weights_path, weights_only=True, map_location=device))
    model.to(device).eval()
    return model


def shard_range(n_items: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous [lo, hi) share of `n_items` for `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_SIDE_STREAMS: dict = {}


def _side_stream(dev, i):
    key = (dev.index, i)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


def exchange_logits(local: torch.Tensor, n_unique: int, group=None) -> torch.Tensor:
    """The one exchange step of the sharded path: all-gather of per-tile logits.

    `local` is this rank's float32[ceil(n_unique/world), n_cls] block (its first
    hi-lo rows are real, the rest padding); returns float32[n_unique, n_cls] in the
    reference's tile order on every rank.  RCCL (backend "nccl") on GPUs; the same
    code runs on gloo/CPU tensors in the tests."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per_rank = local.shape[0]
    gathered = torch.empty((world * per_rank, local.shape[1]), dtype=local.dtype, device=local.device)
    try:
        dist.all_gather_into_tensor(gathered, local.contiguous(), group=group)
    except NotImplementedError:  # a backend without the flat form; a real RCCL failure (RuntimeError) must surface, not be retried
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local.contiguous(), group=group)
        gathered = torch.cat(parts)
    rows = []
    for r in range(world):
        lo, hi = shard_range(n_unique, world, r)
        rows.append(gathered[r * per_rank:r * per_rank + (hi - lo)])
    return torch.cat(rows)


def predict_full_patched(sampler: FullImageDenseSampler, model: ResNet18HIP, n_classes: int,
                         downscale: int = 16, micro_batch: int | None = None, group=None,
                         return_logits: bool = False, streams: int = 2, dedupe_padding: bool = False, timing: list | None = None):
    """Device-resident whole-slide prediction (rows a1-a8 end to end).

    Single process: every tile (padding duplicates included) goes through the fused
    gather+ResNet-18 kernels in micro-batches, logits stay in HBM, one ordered
    accumulate + argmax.  Under torch.distributed (one process per GPU, backend
    "nccl" = RCCL): rank r takes the contiguous range shard_range(n_unique, world, r)
    of the reference-ordered origin list, logits are exchanged with ONE all-gather
    (n_unique x n_cls floats in total), and every rank finishes the map; the corner
    tile's padding duplicates are reconstructed from the gathered logits so the
    canvas equals the single-GPU / reference result.
    `dedupe_padding=True` leaves the padding duplicates of the corner tile out of the accumulation (the reference adds them,
    predict_full_patched.py:49-54, which is the default here).
    `timing`: a list that receives one (start, end) pair of HIP events around the all-gather (bench.py's `allgather_ms`).
    Returns int64[h//d, w//d] on the device (and the float32[n_padded, n_cls] logits).
    """
    import torch.distributed as dist

    streamed = not sampler.resident          # ONDISK_MULTIPROC: row strips are uploaded as they are needed
    slide = None if streamed else sampler.data_device
    dev = sampler.device if streamed else slide.device
    P = sampler.patch_size
    origins = sampler.origins                      # padded, reference order
    n_unique, n_padded = sampler.n_tiles, len(origins)
    # tiles per kernel launch (independent of the sampler's batch size).  bf16: 4 096, the library's maximum (a 64 x 64 x 64-channel
    # map of 4 096 tiles is 2 GiB).  float32: 1 024 -- the same map would be 4 GiB per tensor at 4 096 tiles, past the 32-bit byte
    # offsets of the conv schedule tables (the library refuses it)
    mb = micro_batch or (4096 if getattr(model, "compute_dtype", "f32") == "bf16" else 1024)
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    lo, hi = shard_range(n_unique, world, rank)
    if hi - lo > mb:
        # near-equal launches instead of full ones plus a short tail, each a MULTIPLE OF 128 TILES (all but the last): the persistent
        # kernels run one tile per workgroup per iteration on 256 CUs (stem: 768 workgroups), and a layer has 8 / 4 / 2 / 2 conv tiles
        # per 256 x 256 image, so only multiples of 128 images fill the last iteration of every layer.  38 416 tiles as 10 x 3 842 paid
        # an almost empty extra iteration in every layer of every launch (121 instead of 120.06 in layer 1, 31 instead of 30.02 in
        # layers 3-4, 26 instead of 25.01 in the stem: ~2 % of the slide); 9 x 3 968 + 2 704 does not.
        k = -(-(hi - lo) // mb)
        per = -(-(hi - lo) // k)
        mb = min(mb, max(128, -(-per // 128) * 128)) if mb >= 128 else per
        if os.environ.get("DH_MB_ALIGN") == "0":   # A/B: the round-3 rule (equal launches, any size)
            mb = per
    o_dev = torch.from_numpy(origins[lo:hi]).to(dev)
    per_rank = -(-n_unique // world)
    local = torch.zeros((per_rank, n_classes), dtype=torch.float32, device=dev)
    # parameters are synced to the native handles once; the loop below is launches only.
    # Micro-batches alternate over `streams` HIP streams (one workspace each) so that the short
    # kernels and the tails of one micro-batch overlap with the convolutions of the next.
    handles = model.eval().lane_handles(max(1, streams))
    main = torch.cuda.current_stream(dev)
    lanes = [main] + [_side_stream(dev, i) for i in range(1, len(handles))]
    for st in lanes[1:]:
        st.wait_stream(main)
    fwd = lib().dh_resnet18_forward_tiles
    if streamed:
        _forward_streamed(sampler, handles[0], origins[lo:hi], local, n_classes, mb)
    for k, s in enumerate(range(0, 0 if streamed else hi - lo, mb)):
        e = min(s + mb, hi - lo)
        lane = k % len(handles)
        check(fwd(handles[lane], slide.data_ptr(), sampler.h, sampler.w, o_dev.data_ptr() + 8 * s, e - s, P,
                  local.data_ptr() + 4 * n_classes * s, C.c_void_p(lanes[lane].cuda_stream)),
              "dh_resnet18_forward_tiles")
    for st in lanes[1:]:
        main.wait_stream(st)
    if distributed and timing is not None and local.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record(main)
        logits_unique = exchange_logits(local, n_unique, group)
        ev[1].record(main)
        timing.append(ev)
    else:
        logits_unique = exchange_logits(local, n_unique, group) if distributed else local[:n_unique]
    pad = n_padded - n_unique
    logits = torch.cat([logits_unique, logits_unique[-1:].expand(pad, -1)]) if pad else logits_unique
    if dedupe_padding:
        _, cmap = tiles.accumulate_logits(logits_unique.contiguous(), origins[:n_unique], P, downscale, sampler.h, sampler.w)
    else:
        _, cmap = tiles.accumulate_logits(logits.contiguous(), origins, P, downscale, sampler.h, sampler.w)
    return (cmap, logits) if return_logits else cmap


def _forward_streamed(sampler, handle, origins: np.ndarray, local: torch.Tensor, n_classes: int, micro_batch: int):
    """Logits of `origins` (this rank's range, reference order) when the slide is not resident: the tiles are
    grouped by tile row; the P-row strip of each group is read from the reader into a pinned buffer, uploaded
    on a side stream (two strip buffers: the disk read and the upload of strip k+1 run under the forward of strip k) and
    serves as the 'slide' of dh_resnet18_forward_tiles; logits land at their reference-order positions."""
    dev, P, w = sampler.device, sampler.patch_size, sampler.w
    main = torch.cuda.current_stream(dev)
    copy_stream = torch.cuda.Stream(dev)
    ys = np.unique(origins[:, 0])
    groups = [np.nonzero(origins[:, 0] == y)[0] for y in ys]
    pinned = [torch.empty((P, w, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
    strip = [torch.empty((P, w, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    uploaded = [torch.cuda.Event() for _ in range(2)]
    consumed = [None, None]
    fwd = lib().dh_resnet18_forward_tiles

    def stage(k):
        b = k & 1
        if consumed[b] is not None:
            consumed[b].synchronize()
        y = int(ys[k])
        np.copyto(pinned[b].numpy(), sampler.read_region(y, 0, y + P, w))
        with torch.cuda.stream(copy_stream):
            strip[b].copy_(pinned[b], non_blocking=True)
            uploaded[b].record(copy_stream)

    if len(ys):
        stage(0)
    for k, idx in enumerate(groups):
        b = k & 1
        main.wait_event(uploaded[b])
        o = np.zeros((len(idx), 2), np.int32)
        o[:, 1] = origins[idx, 1]
        o_dev = torch.from_numpy(o).to(dev)
        out = torch.empty((len(idx), n_classes), dtype=torch.float32, device=dev)
        for s0 in range(0, len(idx), micro_batch):
            e0 = min(s0 + micro_batch, len(idx))
            check(fwd(handle, strip[b].data_ptr(), P, w, o_dev.data_ptr() + 8 * s0, e0 - s0, P,
                      out.data_ptr() + 4 * n_classes * s0, C.c_void_p(main.cuda_stream)), "dh_resnet18_forward_tiles")
        local[torch.from_numpy(idx).to(dev)] = out
        consumed[b] = torch.cuda.Event()
        consumed[b].record(main)
        # strip k is queued: NOW read strip k+1 from the reader (the host blocks on the disk while the GPU runs strip k;
        # staging before the launches left the GPU idle during every read)
        if k + 1 < len(ys):
            stage(k + 1)


def perform_and_save_visualizations(img, anno_dsc, pred, out_dir: Path = Path("."), stem: str | None = None,
                                    alpha: float = 0.6, save: bool = True, device="cuda"):
    """Colourised class mask, the slide at the map's resolution and their overlay -- predict_full_patched.py:81-113.

    `img`: path (psimage, when installed: `get_region(..., target_hw)` as the reference) or a uint8[H,W,3]
    array / GPU tensor, which is sampled at the map's resolution by nearest source pixel (psimage's own
    resampler is third-party and unknown here).  The colour lookup and the float64 blend run on the GPU
    (`dh_colorize_map`, `dh_overlay_blend`) and are bit-identical to the reference's NumPy lines.
    Returns (mask, image, overlay) as uint8[h, w, 3] NumPy arrays; JPEGs are written when `save`."""
    dev = torch.device(device)
    pred_t = pred if isinstance(pred, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(pred))
    pred_t = pred_t.to(dev, torch.int64).contiguous()
    h, w = int(pred_t.shape[0]), int(pred_t.shape[1])
    n_ids = max((a.id for a in anno_dsc.anno_classes), default=-1) + 1
    lut = torch.zeros((n_ids, 3), dtype=torch.uint8)
    for a in anno_dsc.anno_classes:
        lut[a.id] = torch.tensor(a.color, dtype=torch.uint8)
    colored = tiles.colorize_map(pred_t, lut)
    if isinstance(img, (str, Path)):
        stem = stem or Path(img).stem
        with open_slide(img) as psim:
            small = torch.from_numpy(np.ascontiguousarray(psim.get_region((0, 0), (psim.height, psim.width), target_hw=(h, w)))).to(dev)
    else:
        full = img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img))
        full = full.to(dev)
        ys = (torch.arange(h, device=dev) * full.shape[0]) // h
        xs = (torch.arange(w, device=dev) * full.shape[1]) // w
        small = full[ys][:, xs].contiguous()
    overlay = tiles.overlay_blend(small, colored, alpha)
    mask_np, img_np, ov_np = colored.cpu().numpy(), small.cpu().numpy(), overlay.cpu().numpy()
    if save:
        from PIL import Image
        out_dir = Path(out_dir)
        out_dir.mkdir(exist_ok=True, parents=True)
        stem = stem or "slide"
        Image.fromarray(mask_np).save(out_dir / f"{stem}_mask.jpg", quality=95)
        Image.fromarray(img_np).save(out_dir / f"{stem}.jpg", quality=95)
        Image.fromarray(ov_np).save(out_dir / f"{stem}_overlay.jpg", quality=95)
    return mask_np, img_np, ov_np


KNOWN_COLORS = {   # predict_full_patched.py:139-148
    "AT": (245, 119, 34),    # orange
    "BG": (153, 255, 255),   # cyan
    "LP": (64, 170, 72),     # green
    "MM": (255, 0, 0),       # red
    "TUM": (33, 67, 156),    # blue
}


def main(argv=None, model=None):
    """The reference's `__main__` (predict_full_patched.py:128-177) as a per-rank program.

    The reference hard-codes the slide path, `./output/best_model.pth`, layer 2, downscale 16, patch 224, batch 64 and
    (dense branch, :165-167) stride 112; those are the defaults of the flags below.  The dense branch is the multi-GPU
    path: under `python -m torch.distributed.run --nproc-per-node N -m examples.predict_full_patched ...` every rank
    binds its GPU, joins the RCCL group, takes its contiguous tile range and the logits are exchanged with one
    all-gather (`predict_full_patched`); rank 0 writes the three JPEGs.  `--random_sampler` keeps the reference's
    default branch (`FullImageRndSampler` through `ImagePredictorPatched.process()`, single process).
    `--synthetic H W` runs on a closed-form slide when no .psi file / psimage is at hand; `--weights ''` = random init.
    `model`: an injected module (tests)."""
    import argparse

    from ..anno.utils import AnnoDescription
    from ..distributed import finalize, init_from_env
    from ..models.patch_cls_simple import utils
    from ..patch_samplers.full_samplers import FullImageRndSampler, SamplerExecutionMode

    ap = argparse.ArgumentParser(description=main.__doc__.splitlines()[0])
    ap.add_argument("--image", default="/home/xubiker/dev/PATH-DT-MSU.WSS2/images/test/test_01.psi")
    ap.add_argument("--synthetic", type=int, nargs=2, metavar=("H", "W"), default=None)
    ap.add_argument("--weights", default="./output/best_model.pth")
    ap.add_argument("--layer", type=int, default=2)
    ap.add_argument("--downscale_vis", type=int, default=16)
    ap.add_argument("--patch_size", type=int, default=224)
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--stride", type=int, default=112)
    ap.add_argument("--random_sampler", action="store_true", default=False)
    ap.add_argument("--ondisk", action="store_true", help="SamplerExecutionMode.ONDISK_MULTIPROC: stream row strips")
    ap.add_argument("--compute_dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--out_dir", default="./output/")
    ap.add_argument("--no_visualizations", action="store_true")
    args = ap.parse_args(argv)

    rank, world, _dev_index, owned = init_from_env()   # binds the rank's GPU before any other GPU call
    ok = False
    try:
        import torch.distributed as dist
        device = utils.get_device()
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if model is None:
            if device.type != "cuda":
                raise RuntimeError("predict_full_patched runs on the GPU only (HIP kernels); no CPU fallback")
            if args.weights:
                model = load_model(args.weights, device, args.compute_dtype)
            else:
                torch.manual_seed(0)   # the same random init on every rank
                model = get_model(n_classes=5, compute_dtype=args.compute_dtype).to(device).eval()
        anno_dsc = AnnoDescription.with_known_colors(KNOWN_COLORS)
        n_cls = len(anno_dsc.anno_classes)
        if args.synthetic is not None:
            img = tiles.synth_slide(args.synthetic[0], args.synthetic[1], 0, device)
            stem = f"synthetic_{args.synthetic[0]}x{args.synthetic[1]}"
        else:
            img, stem = Path(args.image), Path(args.image).stem
        mode = SamplerExecutionMode.ONDISK_MULTIPROC if args.ondisk else SamplerExecutionMode.INMEMORY_SINGLEPROC
        if args.random_sampler:
            if world > 1:
                raise RuntimeError("--random_sampler draws tiles from a running coverage map (one process); "
                                   "the dense sampler is the multi-GPU path")
            smp = FullImageRndSampler(img, layer=args.layer, patch_size=args.patch_size, batch_size=args.batch_size,
                                      mode=mode, device=device)
            pred = ImagePredictorPatched((smp.h, smp.w), patch_sampler=smp.generator(),
                                         batch_predictor=lambda patches: batch_predictor(patches, model, device),
                                         anno=anno_dsc, layer=args.layer, downscale=args.downscale_vis, device=device).process()
        else:
            smp = FullImageDenseSampler(img, layer=args.layer, patch_size=args.patch_size, batch_size=args.batch_size,
                                        mode=mode, stride=args.stride, device=device)
            pred = predict_full_patched(smp, model, n_cls, downscale=args.downscale_vis)   # sharded when world > 1
        if rank == 0 and not args.no_visualizations:
            src = img if isinstance(img, torch.Tensor) or not smp.resident else smp.data_device
            perform_and_save_visualizations(src, anno_dsc, pred, out_dir=Path(args.out_dir), stem=stem, device=device)
        if world > 1:
            dist.barrier()
        ok = True
        return pred
    finally:
        finalize(owned, ok)   # a failing rank leaves without a barrier (distributed.finalize)


if __name__ == "__main__":
    main()
