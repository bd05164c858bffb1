"""Slide sources and the Patch record.

The reference reads slides through the third-party `psimage` package
(`PSImage(path)`, `layer_size`, `get_region_from_layer`) and
wraps tiles in `psimage.core.patches.Patch(layer, pos_x, pos_y, patch_size, data)`.
`psimage` is used when it is importable; otherwise the same duck-typed protocol
is served by in-memory arrays and by memory-mapped `.npy` files, so samplers accept a
path, an array or a reader object.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Any

import numpy as np

try:  # real package, when the user has it
    from psimage.core.patches import Patch  # type: ignore
except Exception:  # pragma: no cover - psimage is absent in the build image

    @dataclass
    class Patch:  # same field order as psimage's record (region_samplers.py:508-512)
        layer: int
        pos_x: int
        pos_y: int
        patch_size: int
        data: Any


class ArraySlide:
    """`PSImage`-shaped reader over an in-memory uint8[h, w, 3] array (single layer)."""

    def __init__(self, array: np.ndarray):
        a = np.asarray(array)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("slide array must be uint8[h, w, 3]")
        self._a = a
        self.height, self.width = a.shape[:2]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def close(self):
        pass

    def _assert_layer(self, layer):
        if layer < 1:
            raise ValueError(f"invalid layer {layer}")

    def layer_size(self, layer):
        return self._a.shape[0], self._a.shape[1]

    def get_region_from_layer(self, layer, p0, p1):
        return self._a[p0[0]:p1[0], p0[1]:p1[1], :]

    def get_region(self, p0, p1, target_hw=None):
        """`PSImage.get_region(p0, p1, target_hw)` as examples/predict_full_patched.py:104 calls it: the region resampled to
        `target_hw`.  psimage's own resampler is third-party and unknown here: nearest source pixel (`floor(i * h / th)`)."""
        reg = self._a[p0[0]:p1[0], p0[1]:p1[1], :]
        if target_hw is None:
            return np.ascontiguousarray(reg)
        th, tw = target_hw
        ys = (np.arange(th) * reg.shape[0]) // th
        xs = (np.arange(tw) * reg.shape[1]) // tw
        return np.ascontiguousarray(reg[ys][:, xs])


def open_slide(source):
    """Return a PSImage-like reader for a path, an ndarray or a reader object."""
    if isinstance(source, np.ndarray):
        return ArraySlide(source)
    if hasattr(source, "layer_size") and hasattr(source, "get_region_from_layer"):
        return source
    if isinstance(source, (str, Path)) and Path(source).suffix == ".npy":
        # a plain on-disk slide: uint8[h, w, 3] in NumPy's .npy container, memory-mapped (never read whole)
        return ArraySlide(np.load(source, mmap_mode="r"))
    if isinstance(source, (str, Path)):
        try:
            from psimage.core.image import PSImage  # type: ignore
        except Exception as e:
            raise ImportError(
                f"reading '{source}' needs the third-party `psimage` package (absent here); "
                "pass a uint8[h,w,3] array, a GPU tensor or a PSImage-like reader instead") from e
        return PSImage(source)
    raise TypeError(f"cannot open a slide from {type(source).__name__}")
