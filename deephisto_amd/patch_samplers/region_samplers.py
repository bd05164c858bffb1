"""Annotated-region samplers -- the a9 boundary of patch_samplers/region_samplers.py.

The reference's `AnnoRegionRndSampler` (region_samplers.py:252-796) picks an annotated
polygon with area-dependent weights, then random patch origins whose overlap with the
polygon is >= `region_intersection`, reads them from the `.psi` file in worker processes and
yields, from `torch_generator` (region_samplers.py:641-738):
    features float32[B, P, P, 3] = uint8 / 255  (then `transforms(features)`),
    labels   int64[B]   (index of the class in the sorted class list),
    coords   float32[B, 2] = (pos_y, pos_x).
That OUTPUT CONTRACT is the hot-path boundary.  Two producers keep it:

* `RectRegionRndSampler` (synthetic data, BASELINE configs[1]): axis-aligned rectangles, for which the
  overlap constraint is closed form; origins from a seeded NumPy generator;
* `AnnoRegionRndSampler` / `AnnoRegionDenseSampler` / `RegionAnnotation` (second half of this file;
  SURVEY section 8f row 2): the reference's polygon annotations, weights and random stream, with the
  shapely geometry restated in `polygon.py`.

In both, pixels never leave HBM: batches are cut, normalised (k/255), laid out and flipped by
`dh_tile_gather` / `dh_tile_gather_aug`.
"""
from __future__ import annotations

import json
from collections import defaultdict
from dataclasses import dataclass
from pathlib import Path
from typing import Callable, Iterator, Sequence

import numpy as np
import torch

from . import polygon as _pg
from .. import tiles
from .._lib import DH_LAYOUT_NCHW, DH_LAYOUT_NHWC
from ..psimage_compat import Patch, open_slide


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
        up = getattr(self, "_uploader", None)
        if up is None:
            up = self._uploader = tiles.PinnedUploader(dev)   # origins / labels through pinned staging: the host never waits for the GPU
        for _ in range(n_batches):
            yx, lab = self.sample_origins(batch_size)
            fh = bool(flips and self._rng.random() < 0.5)
            fv = bool(flips and self._rng.random() < 0.5)
            o_dev, lab_dev, coords = up.upload_batch(yx, lab)   # one staged copy per batch (origins, labels, float coordinates)
            x = tiles.gather_tiles_aug(self.slide, o_dev, self.patch_size, DH_LAYOUT_NCHW, dtype, fh, fv)
            yield x, lab_dev, coords


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


# ======================================================================================================
# Polygon annotations: RegionAnnotation / AnnoRegionRndSampler / AnnoRegionDenseSampler
# (patch_samplers/region_samplers.py:19-871 of the reference)
# ======================================================================================================
# Host producers with the reference's constructor arguments, attributes, weights and -- where the
# reference draws random numbers -- the same calls to the GLOBAL NumPy RNG in the same order.  The
# geometry that the reference takes from shapely is restated in `polygon.py` (parity unpinned: no
# fixture exists and shapely is absent).  What moves to the GPU is, as everywhere in this package,
# the pixel work: slides are resident in HBM and batches are cut / normalised / laid out / flipped by
# the gather kernels; there are no reader worker processes (`batches_per_worker` / `max_workers` only
# shape the chunking of the random stream, as in the reference).
class RegionAnnotation:
    """One annotated polygon (region_samplers.py:19-191).  `vertices` are float64[N, 2] (x, y) in layer-1
    coordinates; `polygon` holds the counter-clockwise ring at the sampler's layer scale."""

    def __init__(self, img_path, region_idx: int, class_: str, vertices: np.ndarray, layer: int,
                 layer_size: tuple[int, int]):
        self.file_path = img_path
        self.region_idx = region_idx
        self.class_ = class_
        self.vertices = vertices
        self._layer = layer
        self._layer_size = layer_size
        if len(vertices.shape) != 2 or vertices.shape[1] != 2:
            raise RuntimeError("Invalid region shape. It should be (N, 2).")
        if vertices.dtype != np.float64:
            raise RuntimeError("Invalid region dtype. It should be float64.")
        ring = _pg.as_ccw(_pg.drop_repeats(vertices if layer == 1 else vertices.copy() / layer))
        if len(ring) >= 3 and _pg.is_simple(ring) and _pg.area(ring) != 0.0:
            self.polygon = ring           # one counter-clockwise simple ring
            self.area = _pg.area(ring)
            self.bounds = _pg.bounds(ring)   # (minx, miny, maxx, maxy), shapely's order
        else:
            # "invalid polygon found. Fixing..." (:68-71): shapely's buffer(0), restated in polygon.repair -- the lobes of the
            # ring that are wound like the ring itself; the region becomes a LIST of simple rings
            print("invalid polygon found. Fixing...")
            rings = _pg.repair(ring) if len(ring) >= 3 else []
            if not rings:
                raise RuntimeError("Invalid (degenerate) polygon.")
            self.polygon = rings if len(rings) > 1 else rings[0]
            self.area = float(sum(_pg.area(r) for r in rings))
            allv = np.concatenate(rings)
            self.bounds = _pg.bounds(allv)

    def __str__(self) -> str:
        stem = Path(self.file_path).stem if isinstance(self.file_path, (str, Path)) else "array"
        return f"Region [{stem}, {self.region_idx}, {self.class_}, {self.vertices.shape}, {round(self.area, 0)}]"

    def _extract_patch_coords_rnd(self, patch_size: int, n_patches: int, region_intersection: float = 0.75,
                                  miss_limit: int = 500) -> list[tuple[int, int]]:
        """Random (y, x) origins whose patch overlaps the polygon by more than `region_intersection` of its
        area -- region_samplers.py:82-143: one `np.random.randint` for x, then one for y, per attempt."""
        ps = patch_size
        h, w = self._layer_size
        x0, y0, x1, y1 = self.bounds
        if self.area < ps * ps * region_intersection:
            raise RuntimeError("Region is too small.")
        coords = []
        for _ in range(n_patches):
            n_miss = 0
            while n_miss < miss_limit:
                x = np.random.randint(x0, min(max(x0 + 1, x1 - ps), w))
                y = np.random.randint(y0, min(max(y0 + 1, y1 - ps), h))
                if float(_pg.overlap_area_square(self.polygon, x, y, ps)) > ps * ps * region_intersection:
                    coords.append((y, x))
                    break
                n_miss += 1
            if n_miss >= miss_limit:
                raise RuntimeError("Miss limit reached. Probably region is too small.")
        return coords

    def _extract_patch_coords_dense(self, patch_size: int, stride: int, region_intersection: float = 0.75
                                    ) -> list[tuple[int, int]]:
        """Grid origins inside the bounding box with enough overlap (region_samplers.py:145-191), row-major."""
        h, w = self._layer_size
        x0, y0, x1, y1 = (round(v) for v in self.bounds)
        x1, y1 = min(x1, w - patch_size), min(y1, h - patch_size)
        ys, xs = np.arange(y0, y1, stride), np.arange(x0, x1, stride)
        if len(ys) == 0 or len(xs) == 0:
            return []
        yy, xx = np.meshgrid(ys, xs, indexing="ij")
        ia = _pg.overlap_area_square(self.polygon, xx.ravel(), yy.ravel(), patch_size)
        keep = ia > patch_size * patch_size * region_intersection
        return [(int(y), int(x)) for y, x in zip(yy.ravel()[keep], xx.ravel()[keep])]


def _load_annotation(anno):
    if isinstance(anno, (str, Path)):
        with open(anno) as f:
            return json.load(f)
    return anno   # already the list of {"class": ..., "vertices": [[x, y], ...]} records


def _parse_annotations(img_anno_paths, layer: int, classes: list[str] | None = None):
    """region_samplers.py:194-249: (regions of every class over all images, the same per image)."""
    regions_all = defaultdict(list)
    regions_per_image = [defaultdict(list) for _ in img_anno_paths]
    regions_failed = 0
    for j, (img, anno) in enumerate(img_anno_paths):
        with open_slide(img) as psim:
            size = psim.layer_size(layer)
        for i, a in enumerate(_load_annotation(anno)):
            cls = a["class"]
            if classes is not None and cls not in classes:
                continue
            try:
                reg = RegionAnnotation(img_path=img, region_idx=i, class_=cls,
                                       vertices=np.array(a["vertices"], dtype=np.float64), layer=layer, layer_size=size)
                reg.image_index = j
                regions_per_image[j][cls].append(reg)
                regions_all[cls].append(reg)
            except Exception:
                regions_failed += 1
    if regions_failed > 0:
        print(f"Failed to parse {regions_failed} regions.")
    print(f"regions all: { {cls: len(r) for cls, r in regions_all.items()} }")
    return regions_all, regions_per_image


class _SlideBank:
    """The images of a sampler: PSImage-like host readers + their layers resident in HBM (uploaded on first use)."""

    def __init__(self, sources, layer: int, device):
        self._readers = [open_slide(s) if not isinstance(s, torch.Tensor) else None for s in sources]
        self._dev = [s.to(device).contiguous() if isinstance(s, torch.Tensor) else None for s in sources]
        self.layer, self.device = layer, torch.device(device)

    def size(self, j: int) -> tuple[int, int]:
        if self._readers[j] is None:
            return int(self._dev[j].shape[0]), int(self._dev[j].shape[1])
        return tuple(self._readers[j].layer_size(self.layer))

    def host_patch(self, j: int, y: int, x: int, ps: int):
        """uint8[ps, ps, 3] at (y, x).  The reference's origin bounds (region_samplers.py:112-118, 160-166)
        let a patch hang over the image border; what psimage returns there is outside this repository, here
        the pixels outside the image are 0 -- the same rule as the device gather (dh_tile_gather_aug)."""
        h, w = self.size(j)
        ya, yb, xa, xb = max(y, 0), min(y + ps, h), max(x, 0), min(x + ps, w)
        inside = ya == y and xa == x and yb == y + ps and xb == x + ps
        if yb <= ya or xb <= xa:
            return np.zeros((ps, ps, 3), np.uint8)
        if self._readers[j] is None:
            part = self._dev[j][ya:yb, xa:xb, :].cpu().numpy()
        else:
            part = np.asarray(self._readers[j].get_region_from_layer(self.layer, (ya, xa), (yb, xb)))
        if inside:
            return part
        out = np.zeros((ps, ps, 3), np.uint8)
        out[ya - y:yb - y, xa - x:xb - x, :] = part
        return out

    def slide(self, j: int) -> torch.Tensor:
        if self._dev[j] is None:
            r = self._readers[j]
            h, w = r.layer_size(self.layer)
            self._dev[j] = torch.from_numpy(np.ascontiguousarray(r.get_region_from_layer(self.layer, (0, 0), (h, w)))).to(self.device)
        return self._dev[j]


class AnnoRegionRndSampler:
    """Area-weighted random patches from annotated polygons -- drop-in for region_samplers.py:252-796.

    `img_anno_paths` pairs an image (path to a psimage file when that package is installed, a uint8[h,w,3]
    array / GPU tensor, or any PSImage-like reader) with its annotation (path to the JSON list of
    `{"class", "vertices"}` records, or that list)."""

    def __init__(self, img_anno_paths, layer: int, patch_size: int, region_intersection: float = 0.75,
                 patches_from_one_region: int = 4, region_area_influence: float = 0.5, classes: list[str] = None,
                 one_image_for_batch: bool = False, device="cuda"):
        self.img_anno_paths = img_anno_paths
        self.layer = layer
        self.patch_size = patch_size
        self.region_intersection = region_intersection
        self.patches_from_one_region = patches_from_one_region
        self.region_area_influence = region_area_influence
        self.one_image_for_batch = one_image_for_batch
        self.regions, self.regions_per_image = _parse_annotations(img_anno_paths, layer=layer, classes=classes)
        self.classes = sorted(list(self.regions.keys()))
        if not self.classes:
            raise ValueError("no usable annotated regions")
        self._reg_w_all, self._reg_w_per_img, self._img_w, self._img_w_all = self._calc_weights(
            self.regions, self.regions_per_image)
        self._bank = _SlideBank([p[0] for p in img_anno_paths], layer, device)

    # ---- weights (region_samplers.py:339-482) ------------------------------------------------------------
    def _calc_area_weights(self, areas, area_influence: float):
        """Mixing weights of regions (or images) from their areas (region_samplers.py:339-378): uniform at influence 0, pulled
        towards the area shares for positive influence and towards the inverse-area shares for negative influence, renormalised.
        The sums are Python's left-to-right `sum`, as in the reference, so the weights are the same floats."""
        if not -1 <= area_influence <= 1:
            raise AssertionError("area influence must lie in [-1, 1]")
        n = len(areas)
        uniform = np.full(n, 1.0, dtype=np.float64) / n
        if area_influence == 0:
            return uniform
        pull = list(areas) if area_influence > 0 else [1 / a for a in areas]
        share = np.array(pull) / sum(pull)
        mixed = uniform + (share - uniform) * abs(area_influence)
        return mixed / sum(mixed)

    def _calc_weights(self, regions, regions_per_image):
        infl = self.region_area_influence
        reg_weights_all = {cls: self._calc_area_weights([r.area for r in reg], infl) for cls, reg in regions.items()}
        reg_weights_per_img = [{cls: self._calc_area_weights([r.area for r in reg], infl) for cls, reg in rpi.items()}
                               for rpi in regions_per_image]
        img_weights = {}
        for cls in self.classes:
            a = np.array([sum(r.area for r in (rpi[cls] if cls in rpi else [])) for rpi in regions_per_image])
            img_weights[cls] = a / np.sum(a)
        all_areas = [sum(sum(j.area for j in i) for i in rpi.values()) for rpi in regions_per_image]
        img_weights_all = self._calc_area_weights(all_areas, infl)
        return reg_weights_all, reg_weights_per_img, img_weights, img_weights_all

    def __len__(self):   # region_samplers.py:788-796
        ps = self.patch_size * self.layer
        return int(sum(sum(r.area for r in lst) for lst in self.regions.values()) / (ps * ps))

    # ---- the random stream (region_samplers.py:525-591) --------------------------------------------------
    def _records(self, n: int, cls_idx: int = None) -> list[tuple[int, int, int, int]]:
        """n (image index, y, x, class index) records, drawn like `_gen_single_proc`: class, region
        (weighted), then `patches_from_one_region` origins from that region; a region that cannot deliver
        (too small, miss limit) is skipped and another draw made, as the reference's `except: continue`."""
        res = []
        if self.one_image_for_batch:
            img_idx = int(np.random.choice(len(self.img_anno_paths), p=self._img_w_all))
            classes_for_img = self._reg_w_per_img[img_idx].keys()
            classes_idx = [self.classes.index(cls) for cls in classes_for_img]
        while len(res) < n:
            try:
                if self.one_image_for_batch:
                    c_idx = cls_idx or np.random.choice(classes_idx)          # `or`: class 0 cannot be forced (reference, :552)
                    cls = self.classes[c_idx]
                    if cls not in classes_for_img:
                        raise RuntimeError(f"Class {cls} not found in image")
                    regs, w = self.regions_per_image[img_idx][cls], self._reg_w_per_img[img_idx][cls]
                else:
                    c_idx = cls_idx or np.random.randint(len(self.classes))  # same `or` (:572)
                    cls = self.classes[c_idx]
                    regs, w = self.regions[cls], self._reg_w_all[cls]
                region = regs[int(np.random.choice(len(regs), p=w))]
                k = min(self.patches_from_one_region, n - len(res))
                coords = region._extract_patch_coords_rnd(n_patches=k, patch_size=self.patch_size,
                                                          region_intersection=self.region_intersection)
                res.extend((region.image_index, int(y), int(x), int(c_idx)) for y, x in coords)
            except RuntimeError:
                continue
        return res

    def _split_chunks(self, n, k):
        q = [k] * (n // k)
        if n % k > 0:
            q.append(n % k)
        return q

    def _gen_single_proc(self, n: int, cls_idx: int = None) -> list[tuple[Patch, int]]:
        ps = self.patch_size
        return [(Patch(self.layer, pos_x=x, pos_y=y, patch_size=ps, data=self._bank.host_patch(j, y, x, ps)), c)
                for j, y, x, c in self._records(n, cls_idx)]

    def structs_generator(self, batch_size: int, n_batches: int, batches_per_worker: int = 2, max_workers: int = None,
                          cls_idx: int = None) -> Iterator[list[tuple[Patch, int]]]:
        """Lists of `batch_size` (Patch, class index) pairs (region_samplers.py:641-683)."""
        for i in self._split_chunks(n_batches, batches_per_worker):
            lst = self._gen_single_proc(batch_size * i, cls_idx)
            for k in range(0, len(lst), batch_size):
                yield lst[k:k + batch_size]

    # ---- device batches --------------------------------------------------------------------------------
    def _assemble(self, recs, layout: int, dtype, flip_h: bool = False, flip_v: bool = False):
        """Cut the records' patches from the HBM-resident slides into one batch tensor (+ labels, coords)."""
        dev, ps = self._bank.device, self.patch_size
        arr = np.array(recs, dtype=np.int64).reshape(-1, 4)
        out = None
        up = getattr(self, "_uploader", None)
        if up is None:
            up = self._uploader = tiles.PinnedUploader(dev, depth=8)   # non-blocking uploads (tiles.PinnedUploader)
        if (arr[:, 0] == arr[0, 0]).all():   # the whole batch from one slide (always so with one_image_for_batch): one staged copy
            o_dev, labels, coords = up.upload_batch(arr[:, 1:3].astype(np.int32), arr[:, 3].copy())
            return tiles.gather_tiles_aug(self._bank.slide(int(arr[0, 0])), o_dev, ps, layout, dtype, flip_h, flip_v), labels, coords
        for j in np.unique(arr[:, 0]):
            sel = np.nonzero(arr[:, 0] == j)[0]
            o_dev = up.upload(arr[sel, 1:3].astype(np.int32))
            part = tiles.gather_tiles_aug(self._bank.slide(int(j)), o_dev, ps, layout, dtype, flip_h, flip_v)
            if len(sel) == len(arr):
                out = part
            else:
                if out is None:
                    out = torch.empty((len(arr),) + tuple(part.shape[1:]), dtype=part.dtype, device=dev)
                out[up.upload(sel.astype(np.int64))] = part
        labels = up.upload(arr[:, 3].copy())
        coords = up.upload(arr[:, 1:3].astype(np.float32))
        return out, labels, coords

    def torch_generator(self, batch_size: int, n_batches: int, batches_per_worker: int = 2,
                        transforms: Callable | None = None, max_workers: int = None, cls_idx: int = None
                        ) -> Iterator[tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """(features f32[B,P,P,3] = uint8/255, labels int64[B], coords f32[B,2] (y, x)) on the GPU, then
        `transforms(features)` -- region_samplers.py:685-738.  As in the reference, `cls_idx` is accepted but
        not forwarded (`_gen_single_proc_torch(n)`, :725), and one chunk of `batches_per_worker` batches is one
        run of the random stream (so `one_image_for_batch` picks its image once per chunk)."""
        for i in self._split_chunks(n_batches, batches_per_worker):
            recs = self._records(batch_size * i)
            for k in range(0, len(recs), batch_size):
                features, labels, coords = self._assemble(recs[k:k + batch_size], DH_LAYOUT_NHWC, torch.float32)
                if transforms is not None:
                    features = transforms(features)
                yield features, labels, coords

    def device_batches(self, batch_size: int, n_batches: int, flips: bool = True, dtype=torch.float32,
                       batches_per_worker: int = 2) -> Iterator[tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """Training fast path: [B,3,P,P] batches with train.py:69-81's permute + RandomHorizontalFlip +
        RandomVerticalFlip (one torch coin each per batch, in that order) fused into the gather kernel."""
        for i in self._split_chunks(n_batches, batches_per_worker):
            recs = self._records(batch_size * i)
            for k in range(0, len(recs), batch_size):
                fh = bool(flips and torch.rand(1).item() < 0.5)
                fv = bool(flips and torch.rand(1).item() < 0.5)
                yield self._assemble(recs[k:k + batch_size], DH_LAYOUT_NCHW, dtype, fh, fv)


    def torch_iterable_dataset(self):
        """Endless `IterableDataset` of single (features f32[P,P,3] = uint8/255, label int64, coords f32[2]) samples
        for use with a DataLoader (region_samplers.py:740-786).  Samples are cut on the GPU
        `patches_from_one_region` at a time; coords are (pos_y, pos_x) -- the reference's inner generator
        writes pos_y twice (:770-772), which looks unintended and is not reproduced."""
        from torch.utils.data import IterableDataset

        sampler = self

        class _Dataset(IterableDataset):
            def __iter__(self):
                while True:
                    recs = sampler._records(sampler.patches_from_one_region)
                    f, lab, c = sampler._assemble(recs, DH_LAYOUT_NHWC, torch.float32)
                    for i in range(len(recs)):
                        yield f[i], lab[i], c[i]

        return _Dataset()


class AnnoRegionDenseSampler:
    """Every grid patch of every annotated region, class by class (region_samplers.py:799-871)."""

    def __init__(self, img_anno_paths, layer: int, patch_size: int, stride: int, region_intersection: float = 0.75,
                 classes: list[str] = None, device="cuda"):
        self.img_anno_paths = img_anno_paths
        self.layer = layer
        self.patch_size = patch_size
        self.stride = stride
        self.region_intersection = region_intersection
        self.regions, _ = _parse_annotations(img_anno_paths, layer=layer, classes=classes)
        self.classes = sorted(list(self.regions.keys()))
        self._bank = _SlideBank([p[0] for p in img_anno_paths], layer, device)

    def _patches_one_region(self, region: RegionAnnotation) -> list[Patch]:
        ps = self.patch_size
        coords = region._extract_patch_coords_dense(patch_size=ps, stride=self.stride,
                                                    region_intersection=self.region_intersection)
        return [Patch(self.layer, pos_x=c[1], pos_y=c[0], patch_size=ps,
                      data=self._bank.host_patch(region.image_index, c[0], c[1], ps)) for c in coords]

    def structs_generator(self) -> Iterator[tuple[Patch, int]]:
        for cls_idx, cls in enumerate(self.classes):
            for region in self.regions[cls]:
                for p in self._patches_one_region(region):
                    yield p, cls_idx

    def device_batches(self, batch_size: int, layout: int = DH_LAYOUT_NCHW, dtype=torch.float32
                       ) -> Iterator[tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """The same patches in the same order as device tensors (tiles cut on the GPU), `batch_size` at a time."""
        dev, ps = self._bank.device, self.patch_size
        for cls_idx, cls in enumerate(self.classes):
            for region in self.regions[cls]:
                coords = region._extract_patch_coords_dense(patch_size=ps, stride=self.stride,
                                                            region_intersection=self.region_intersection)
                for k in range(0, len(coords), batch_size):
                    o = np.array(coords[k:k + batch_size], dtype=np.int32)
                    o_dev = torch.from_numpy(o).to(dev)
                    # bounds-safe gather (zero outside the image): dense origins of a region that touches or leaves
                    # the image border can be negative or hang over it (region_samplers.py:160-166)
                    x = tiles.gather_tiles_aug(self._bank.slide(region.image_index), o_dev, ps, layout, dtype)
                    yield x, torch.full((len(o),), cls_idx, dtype=torch.int64, device=dev), tiles.tile_coords(o_dev)


def extract_and_save_subset(img_anno_paths, out_folder, patch_size: int, layer: int, patches_per_class: int,
                            intersection: float = 0.95, device="cuda"):
    """`patches_per_class` JPEG patches per class under `out_folder/<class index>/<n>.jpg` -- the test ImageFolder of
    models/patch_cls_simple/train.py:41-56 (region_samplers.py:874-909): regions weighted equally, one patch per region,
    95 % of a patch inside its region, batches of 4.  As in the reference, class index 0 cannot be forced (`cls_idx or random`,
    region_samplers.py:555, 576): folder "0" receives patches of randomly drawn classes."""
    from pathlib import Path as _Path

    from PIL import Image

    sampler = AnnoRegionRndSampler(img_anno_paths=img_anno_paths, layer=layer, patch_size=patch_size,
                                   region_intersection=intersection, region_area_influence=0, patches_from_one_region=1,
                                   device=device)
    out_folder = _Path(out_folder)
    batch_size = 4
    counts = {}
    for cls_idx, cls in enumerate(sampler.classes):
        (out_folder / str(cls_idx)).mkdir(parents=True, exist_ok=True)
        n = patches_per_class // batch_size
        count = 0
        for batch in sampler.structs_generator(batch_size=batch_size, n_batches=n, cls_idx=cls_idx):
            for patch, _cls in batch:
                Image.fromarray(np.ascontiguousarray(patch.data)).save(out_folder / str(cls_idx) / f"{count}.jpg")
                count += 1
        counts[cls] = count
    return counts

