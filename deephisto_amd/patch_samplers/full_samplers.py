"""Full-slide samplers -- drop-in for patch_samplers/full_samplers.py (dense path).

`FullImageDenseSampler` keeps the reference's constructor, attributes and iterator
protocol (full_samplers.py:302-452).  What changes is where the work happens:

* the tile-origin list is produced by `dh_tile_grid` (bit-exact with
  `_create_batched_coords`, full_samplers.py:374-404);
* the slide layer is made resident in HBM once (uint8 HWC), and
  `generator_torch()` / `generator_device()` cut, normalise (k/255) and lay out the
  batches with the `dh_tile_gather` kernel instead of NumPy slicing + astype +
  divide + torch.tensor (full_samplers.py:353-369, 441-443);
* `generator()` still yields `list[Patch]` for callers that bring their own
  `batch_predictor`; `Patch.data` is a lazy host view so nothing is copied unless a
  caller actually reads the pixels on the CPU.

Documented deviations from the reference:
  `stride=None` means `stride = patch_size` (the reference crashes);
  a slide smaller than the patch raises ValueError (the reference yields negative origins);
  `mode` defaults to INMEMORY_SINGLEPROC (predict_full_patched.py:165-167 omits it);
  tensors from `generator_torch()` are on the sampler's device, not on the CPU;
  ONDISK_MULTIPROC keeps ONE reader open and reads patches / row strips on demand (pinned staging
  buffers, upload on a side stream) instead of a pool of worker processes that re-open the file per
  batch (:332-351, 406-423); read errors propagate instead of being printed and the batch dropped.
"""
from __future__ import annotations

from enum import Enum
from pathlib import Path
from typing import Iterable, Iterator

import numpy as np
import torch

from .. import tiles
from .._lib import DH_LAYOUT_NCHW, DH_LAYOUT_NHWC
from ..psimage_compat import Patch, open_slide


class SamplerExecutionMode(Enum):  # full_samplers.py:16-18
    INMEMORY_SINGLEPROC = 1
    ONDISK_MULTIPROC = 2


class DevicePatch(Patch):
    """A `Patch` whose pixels live in the sampler's HBM-resident slide.

    `.data` materialises the uint8[P, P, 3] host view on first access (the reference
    hands out NumPy views of the in-RAM layer, full_samplers.py:361-365)."""

    def __init__(self, layer, pos_x, pos_y, patch_size, sampler):
        self.layer, self.pos_x, self.pos_y, self.patch_size = layer, pos_x, pos_y, patch_size
        self._sampler = sampler

    @property
    def data(self):
        s, P = self._sampler, self.patch_size
        return s.read_region(self.pos_y, self.pos_x, self.pos_y + P, self.pos_x + P)

    def __repr__(self):
        return f"DevicePatch(layer={self.layer}, pos_x={self.pos_x}, pos_y={self.pos_y}, patch_size={self.patch_size})"


class _SlideHolder:
    """Shared slide residency (host array / HBM tensor) of the full-slide samplers."""

    def _open(self, psimage_path, layer, mode, device):
        self._psim_path = psimage_path
        self.mode = mode
        self.layer = layer
        self.device = torch.device(device)
        self._host = None
        self._dev = None
        if isinstance(psimage_path, torch.Tensor):  # already a uint8[h,w,3] tensor (host or HBM)
            t = psimage_path
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise ValueError("slide tensor must be uint8[h, w, 3]")
            self.h, self.w = int(t.shape[0]), int(t.shape[1])
            if t.is_cuda:
                self._dev, self.device = t.contiguous(), t.device
            else:
                self._host = t.contiguous().numpy()
        elif self.mode == SamplerExecutionMode.ONDISK_MULTIPROC:
            # the layer stays where it is (full_samplers.py:332-351 re-opens the file per batch in worker
            # processes); here ONE reader stays open and is read in patches / row strips on demand
            self._reader = open_slide(psimage_path)
            self._reader._assert_layer(layer)
            self.h, self.w = self._reader.layer_size(self.layer)
        else:
            with open_slide(psimage_path) as psim:
                psim._assert_layer(layer)
                self.h, self.w = psim.layer_size(self.layer)
                self._host = np.ascontiguousarray(
                    psim.get_region_from_layer(self.layer, (0, 0), (self.h, self.w)))

    @property
    def resident(self) -> bool:
        """False in ONDISK_MULTIPROC mode: pixels are streamed from the reader, never held whole."""
        return getattr(self, "_reader", None) is None

    def read_region(self, y0: int, x0: int, y1: int, x1: int) -> np.ndarray:
        """uint8[y1-y0, x1-x0, 3] of the layer from wherever it lives (host array, reader, or HBM)."""
        if getattr(self, "_reader", None) is not None:
            return self._reader.get_region_from_layer(self.layer, (y0, x0), (y1, x1))
        return self.data[y0:y1, x0:x1, :]

    @property
    def data(self) -> np.ndarray:
        """uint8[h, w, 3] host array of the layer (reference attribute, :319-320; absent in on-disk mode)."""
        if not self.resident:
            raise AttributeError("ONDISK_MULTIPROC holds no whole-layer array; use read_region()")
        if self._host is None:
            self._host = self._dev.cpu().numpy()
        return self._host

    @property
    def data_device(self) -> torch.Tensor:
        """uint8[h, w, 3] slide resident in HBM (uploaded once, on first use)."""
        if not self.resident:
            raise AttributeError("ONDISK_MULTIPROC streams the slide; nothing is resident in HBM")
        if self._dev is None:
            self._dev = torch.from_numpy(self._host).to(self.device)
        return self._dev


class FullImageRndSampler(_SlideHolder):
    """Coverage-driven random tile sampler -- drop-in for full_samplers.py:21-299.

    Keeps a `dh x dw` hit-count map at `speedup`x downscale and draws `batch_size` tile origins
    per batch from the cells hit fewer than `dense_level` times, until every cell was hit.  The
    index logic runs on the host and consumes the GLOBAL NumPy RNG with the reference's calls in
    the reference's order (so `np.random.seed(s)` reproduces the reference's origins); pixels
    stay in HBM (`DevicePatch`, `dh_tile_gather_raw`).  `generator_torch` yields the raw 0..255
    floats like the reference (:286 has no /255)."""

    def __init__(self, psimage_path, layer: int, patch_size: int, batch_size: int,
                 mode: SamplerExecutionMode = SamplerExecutionMode.INMEMORY_SINGLEPROC,
                 dense_level: int = 2, speedup: int = 16, device="cuda"):
        self._open(psimage_path, layer, mode, device)
        self.dh, self.dw = self.h // speedup, self.w // speedup
        print(f"Image {self.h} x {self.w} at {speedup}x -> {self.dh} x {self.dw}")
        self.patch_size = int(patch_size)
        self.batch_size = int(batch_size)
        self._downscale = int(speedup)
        self.dense_level = dense_level
        self._filled_ratio = []
        self._accum = None
        if self.h < self.patch_size or self.w < self.patch_size:
            raise ValueError(f"slide {self.h}x{self.w} is smaller than the patch {self.patch_size}")

    def _calc_probmap(self):
        p = np.where(self._accum >= self.dense_level, 0, 1)
        if np.count_nonzero(p) < self.batch_size:
            while np.count_nonzero(p) < self.batch_size:
                p[np.random.randint(0, p.shape[0], size=1), np.random.randint(0, p.shape[1], size=1)] = 1
        return p / np.sum(p)

    def _prepare_indices(self, probmap):
        d, P = self._downscale, self.patch_size
        idx = list(np.random.choice(self.dh * self.dw, size=self.batch_size, replace=False, p=probmap.flatten()))
        pd2 = P // d // 2
        out = []
        for ind in idx:  # y jitter first, then x, one randint each (reference order)
            y = (ind // self.dw - pd2) * d + np.random.randint(d)
            x = (ind % self.dw - pd2) * d + np.random.randint(d)
            out.append((max(min(y, self.h - P), 0), max(min(x, self.w - P), 0)))
        return out

    def _update_accum(self, origins):
        d, s = self._downscale, self.patch_size
        for y, x in origins:
            self._accum[y // d:(y + s) // d, x // d:(x + s) // d] += 1
        return np.count_nonzero(self._accum) / self._accum.size

    def _origin_batches(self):
        self._accum = np.zeros([self.dh, self.dw], dtype=np.float32)
        filled = 0
        while filled < 1:
            origins = self._prepare_indices(self._calc_probmap())
            filled = self._update_accum(origins)
            self._filled_ratio.append(filled)
            yield origins, filled

    def __iter__(self) -> Iterator[tuple[list[Patch], float]]:
        return self.generator()

    def generator(self) -> Iterator[tuple[list[Patch], float]]:
        for origins, filled in self._origin_batches():
            yield [DevicePatch(self.layer, int(x), int(y), self.patch_size, self) for y, x in origins], filled

    def generator_torch(self) -> Iterator[tuple[torch.Tensor, torch.Tensor, float]]:
        """(features f32[B,P,P,3] with the RAW 0..255 values -- no /255 here, full_samplers.py:286 --,
        coords f32[B,2] (y,x), filled ratio)."""
        P = self.patch_size
        if not self.resident:
            # ONDISK_MULTIPROC (full_samplers.py:237-262 reads every patch from the file): the batch's patches go from
            # the open reader into one pinned staging image [B*P, P, 3] and are cut from it on the GPU
            for origins, filled in self._origin_batches():
                B = len(origins)
                pinned = torch.empty((B * P, P, 3), dtype=torch.uint8).pin_memory()
                buf = pinned.numpy()
                for j, (y, x) in enumerate(origins):
                    buf[j * P:(j + 1) * P] = self.read_region(int(y), int(x), int(y) + P, int(x) + P)
                img = pinned.to(self.device, non_blocking=True)
                so = torch.stack([torch.arange(B, dtype=torch.int32) * P, torch.zeros(B, dtype=torch.int32)], 1).to(self.device)
                o_dev = torch.tensor(np.asarray(origins), dtype=torch.int32, device=self.device)
                yield tiles.gather_tiles_raw(img, so, P), tiles.tile_coords(o_dev), filled
            return
        slide = self.data_device
        for origins, filled in self._origin_batches():
            o_dev = torch.tensor(origins, dtype=torch.int32, device=slide.device)
            yield tiles.gather_tiles_raw(slide, o_dev, P), tiles.tile_coords(o_dev), filled


class FullImageDenseSampler(_SlideHolder):
    def __init__(
        self,
        psimage_path,
        layer: int,
        patch_size: int,
        batch_size: int,
        mode: SamplerExecutionMode = SamplerExecutionMode.INMEMORY_SINGLEPROC,
        stride: int = None,
        device="cuda",
    ):
        self._open(psimage_path, layer, mode, device)
        self.patch_size = int(patch_size)
        self.batch_size = int(batch_size)
        self.stride = int(stride) if stride is not None else int(patch_size)
        if self.h < self.patch_size or self.w < self.patch_size:
            raise ValueError(f"slide {self.h}x{self.w} is smaller than the patch {self.patch_size}")
        self._origins, self.n_tiles = tiles.tile_grid(self.h, self.w, self.patch_size, self.stride,
                                                      self.batch_size)
        print(f"Image {self.h} x {self.w}")

    # ---- grid ------------------------------------------------------------------------
    @property
    def origins(self) -> np.ndarray:
        """int32[n_batches*batch_size, 2] (y, x) origins incl. the corner padding (a1)."""
        return self._origins

    def _create_batched_coords(self):
        """list[list[(y, x)]] exactly as the reference returns it (:374-404)."""
        o = self._origins.reshape(-1, self.batch_size, 2)
        return [[(int(y), int(x)) for y, x in b] for b in o]

    def __len__(self):
        return len(self._origins) // self.batch_size

    # ---- iterator protocol -------------------------------------------------------------
    def __iter__(self) -> Iterable[tuple[list[Patch], float]]:
        return self.generator()

    def generator(self) -> Iterable[tuple[list[Patch], float]]:
        nb = len(self)
        o = self._origins.reshape(nb, self.batch_size, 2)
        for i in range(nb):
            patches = [DevicePatch(self.layer, int(x), int(y), self.patch_size, self) for y, x in o[i]]
            yield patches, i / nb

    def _staged_batches(self):
        """On-disk mode: (uint8[B*P, P, 3] device staging image, int32[B,2] staging origins on the device,
        batch index) -- the batch's patches are read from the reader into a pinned buffer and uploaded on a
        side stream while the consumer works on the previous batch (two buffers)."""
        nb, B, P = len(self), self.batch_size, self.patch_size
        dev = self.device
        o = self._origins.reshape(nb, B, 2)
        pinned = [torch.empty((B * P, P, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
        staged = [torch.empty((B * P, P, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        ready = [torch.cuda.Event() for _ in range(2)]
        consumed = [None, None]
        copy_stream = torch.cuda.Stream(dev)
        so = torch.stack([torch.arange(B, dtype=torch.int32) * P, torch.zeros(B, dtype=torch.int32)], 1).to(dev)

        def stage(i):
            k = i & 1
            if consumed[k] is not None:
                consumed[k].synchronize()        # the consumer's kernels on that buffer have finished
            buf = pinned[k].numpy()
            for j, (y, x) in enumerate(o[i]):
                buf[j * P:(j + 1) * P] = self.read_region(int(y), int(x), int(y) + P, int(x) + P)
            with torch.cuda.stream(copy_stream):
                staged[k].copy_(pinned[k], non_blocking=True)
                ready[k].record(copy_stream)

        if nb:
            stage(0)
        for i in range(nb):
            k = i & 1
            torch.cuda.current_stream(dev).wait_event(ready[k])
            yield staged[k], so, i               # the consumer queues its kernels on batch i
            consumed[k] = torch.cuda.Event()
            consumed[k].record(torch.cuda.current_stream(dev))
            if i + 1 < nb:
                stage(i + 1)                     # only now read ahead: the GPU is busy with batch i during the disk read

    def generator_device(self, layout: int = DH_LAYOUT_NCHW, dtype=torch.float32
                         ) -> Iterator[tuple[torch.Tensor, np.ndarray, float]]:
        """(tiles on device in `layout`/`dtype`, int32[B,2] host origins, progress)."""
        nb = len(self)
        if not self.resident:
            o = self._origins.reshape(nb, self.batch_size, 2)
            for img, so, i in self._staged_batches():
                yield tiles.gather_tiles(img, so, self.patch_size, layout, dtype, check_bounds=False), o[i], i / nb
            return
        slide = self.data_device
        o = self._origins.reshape(nb, self.batch_size, 2)
        o_dev = torch.from_numpy(self._origins).to(slide.device).reshape(nb, self.batch_size, 2)
        for i in range(nb):
            t = tiles.gather_tiles(slide, o_dev[i], self.patch_size, layout, dtype, check_bounds=False)
            yield t, o[i], i / nb

    def generator_torch(self) -> Iterator[tuple[torch.Tensor, torch.Tensor, float]]:
        """(features f32[B,P,P,3] in [0,1], coords f32[B,2] (y,x), progress) -- :437-452."""
        nb = len(self)
        if not self.resident:
            o_all = torch.from_numpy(self._origins).to(self.device).reshape(nb, self.batch_size, 2)
            for img, so, i in self._staged_batches():
                f = tiles.gather_tiles(img, so, self.patch_size, DH_LAYOUT_NHWC, torch.float32, check_bounds=False)
                yield f, tiles.tile_coords(o_all[i].contiguous()), i / nb
            return
        slide = self.data_device
        o_dev = torch.from_numpy(self._origins).to(slide.device).reshape(nb, self.batch_size, 2)
        for i in range(nb):
            oi = o_dev[i].contiguous()
            features = tiles.gather_tiles(slide, oi, self.patch_size, DH_LAYOUT_NHWC, torch.float32,
                                          check_bounds=False)
            yield features, tiles.tile_coords(oi), i / nb
