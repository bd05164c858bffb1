"""Annotated-region samplers -- the a9 boundary of patch_samplers/region_samplers.py.

The reference's `AnnoRegionRndSampler` (region_samplers.py:252-796) picks an annotated
polygon with area-dependent weights, then random patch origins whose overlap with the
polygon is >= `region_intersection`, reads them from the `.psi` file in worker processes and
yields, from `torch_generator` (region_samplers.py:641-738):
    features float32[B, P, P, 3] = uint8 / 255  (then `transforms(features)`),
    labels   int64[B]   (index of the class in the sorted class list),
    coords   float32[B, 2] = (pos_y, pos_x).
That OUTPUT CONTRACT is the hot-path boundary and is kept here; the
polygon geometry (shapely, absent) is out of scope this round, so regions are axis-aligned
rectangles, for which the overlap constraint is closed form.  Origins are drawn on the host
with a seeded NumPy generator; pixels never leave HBM: batches are cut, normalised (k/255),
laid out and flipped by `dh_tile_gather` / `dh_tile_gather_aug`.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Iterator, Sequence

import numpy as np
import torch

from .. import tiles
from .._lib import DH_LAYOUT_NCHW, DH_LAYOUT_NHWC


@dataclass
class RectRegion:
    cls: str
    y0: int
    x0: int
    y1: int  # exclusive
    x1: int

    @property
    def area(self) -> int:
        return max(0, self.y1 - self.y0) * max(0, self.x1 - self.x0)


class RectRegionRndSampler:
    """Random patches inside class-labelled rectangles of one HBM-resident slide."""

    def __init__(self, slide, regions: Sequence[RectRegion], layer: int, patch_size: int,
                 region_intersection: float = 0.75, patches_from_one_region: int = 4,
                 region_area_influence: float = 0.5, classes: list[str] | None = None, seed: int = 0, device="cuda"):
        if isinstance(slide, np.ndarray):
            slide = torch.from_numpy(np.ascontiguousarray(slide))
        if slide.dtype != torch.uint8 or slide.dim() != 3 or slide.shape[2] != 3:
            raise ValueError("slide must be uint8[h, w, 3]")
        self.slide = slide.to(device).contiguous()
        self.h, self.w = int(slide.shape[0]), int(slide.shape[1])
        self.layer, self.patch_size = layer, int(patch_size)
        self.region_intersection = float(region_intersection)
        self.patches_from_one_region = int(patches_from_one_region)
        keep = [r for r in regions if (classes is None or r.cls in classes) and r.area > 0]
        if not keep:
            raise ValueError("no usable regions")
        self.regions = keep
        self.classes = sorted({r.cls for r in keep})  # region_samplers.py:303
        # area-dependent region weights: w ~ area ** influence (region_samplers.py:395-482 in spirit)
        a = np.array([r.area for r in keep], dtype=np.float64)
        wts = a ** float(region_area_influence)
        self._weights = wts / wts.sum()
        self._rng = np.random.default_rng(seed)

    def __len__(self):  # region_samplers.py:788-796: area / (patch * layer)^2
        return int(sum(r.area for r in self.regions) / (self.patch_size * self.layer) ** 2)

    def _origin_range(self, r: RectRegion):
        """Origins whose patch overlaps the rectangle by >= region_intersection of the patch area,
        restricted (conservatively) to per-axis overlap >= sqrt(intersection) * P, clamped to the slide."""
        P = self.patch_size
        need = int(np.ceil(np.sqrt(self.region_intersection) * P))
        lo_y, hi_y = r.y0 - (P - need), r.y1 - need
        lo_x, hi_x = r.x0 - (P - need), r.x1 - need
        lo_y, lo_x = max(lo_y, 0), max(lo_x, 0)
        hi_y, hi_x = min(hi_y, self.h - P), min(hi_x, self.w - P)
        if hi_y < lo_y or hi_x < lo_x:  # region smaller than the overlap demand: centre the patch on it
            cy = min(max((r.y0 + r.y1 - P) // 2, 0), self.h - P)
            cx = min(max((r.x0 + r.x1 - P) // 2, 0), self.w - P)
            return cy, cy, cx, cx
        return lo_y, hi_y, lo_x, hi_x

    def sample_origins(self, n: int, cls_idx: int | None = None) -> tuple[np.ndarray, np.ndarray]:
        """(int32[n,2] (y,x) origins, int64[n] labels): weighted region choice, `patches_from_one_region`
        patches per chosen region (region_samplers.py:525-591)."""
        idx = np.arange(len(self.regions))
        w = self._weights
        if cls_idx is not None:
            m = np.array([self.classes.index(r.cls) == cls_idx for r in self.regions])
            idx, w = idx[m], w[m] / w[m].sum()
        yx = np.empty((n, 2), np.int32)
        lab = np.empty(n, np.int64)
        k = 0
        while k < n:
            r = self.regions[int(self._rng.choice(idx, p=w))]
            lo_y, hi_y, lo_x, hi_x = self._origin_range(r)
            take = min(self.patches_from_one_region, n - k)
            yx[k:k + take, 0] = self._rng.integers(lo_y, hi_y + 1, take)
            yx[k:k + take, 1] = self._rng.integers(lo_x, hi_x + 1, take)
            lab[k:k + take] = self.classes.index(r.cls)
            k += take
        return yx, lab

    def torch_generator(self, batch_size: int, n_batches: int, batches_per_worker: int = 2,
                        transforms: Callable | None = None, max_workers: int | None = None,
                        cls_idx: int | None = None) -> Iterator[tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """Same signature and output contract as the reference's torch_generator (:641-738);
        `batches_per_worker` / `max_workers` are accepted and unused (no worker processes:
        the GPU cuts the patches)."""
        dev = self.slide.device
        for _ in range(n_batches):
            yx, lab = self.sample_origins(batch_size, cls_idx)
            o_dev = torch.from_numpy(yx).to(dev)
            features = tiles.gather_tiles(self.slide, o_dev, self.patch_size, DH_LAYOUT_NHWC, torch.float32, check_bounds=False)
            if transforms is not None:
                features = transforms(features)
            yield features, torch.from_numpy(lab).to(dev), tiles.tile_coords(o_dev)

    def device_batches(self, batch_size: int, n_batches: int, flips: bool = True, dtype=torch.float32
                       ) -> Iterator[tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """Fast path of the training loop: [B,3,P,P] batches with train.py:71-81's
        permute + batch-level random flips fused into the gather kernel."""
        dev = self.slide.device
        for _ in range(n_batches):
            yx, lab = self.sample_origins(batch_size)
            fh = bool(flips and self._rng.random() < 0.5)
            fv = bool(flips and self._rng.random() < 0.5)
            o_dev = torch.from_numpy(yx).to(dev)
            x = tiles.gather_tiles_aug(self.slide, o_dev, self.patch_size, DH_LAYOUT_NCHW, dtype, fh, fv)
            yield x, torch.from_numpy(lab).to(dev), tiles.tile_coords(o_dev)


def synthetic_regions(h: int, w: int, n_classes: int = 5, per_class: int = 6, min_side: int = 300,
                      max_side: int = 1200, seed: int = 0) -> list[RectRegion]:
    """Deterministic rectangular 'annotations' for synthetic slides (BASELINE configs[1])."""
    rng = np.random.default_rng(seed)
    names = ["AT", "BG", "LP", "MM", "TUM", "DYS", "C6", "C7"][:n_classes]
    out = []
    for name in names:
        for _ in range(per_class):
            hh = int(rng.integers(min_side, min(max_side, h) + 1))
            ww = int(rng.integers(min_side, min(max_side, w) + 1))
            y0 = int(rng.integers(0, h - hh + 1))
            x0 = int(rng.integers(0, w - ww + 1))
            out.append(RectRegion(name, y0, x0, y0 + hh, x0 + ww))
    return out
