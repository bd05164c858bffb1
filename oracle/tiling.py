"""Oracle for the dense tiling / normalisation / accumulation rows (a1-a5, a8).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  NumPy only.
"""
from __future__ import annotations

import numpy as np


def tile_origins(h: int, w: int, patch: int, stride: int) -> np.ndarray:
    """int32[n, 2] (y, x) tile origins in the reference's order (a1).

    Follows FullImageDenseSampler._create_batched_coords,
    patch_samplers/full_samplers.py:380-397: interior grid (y-major) over
    range(0, h-P, S) x range(0, w-P, S); then the last column (y, w-P); then
    the last row (h-P, x); then the corner (h-P, w-P).
    """
    ys = np.arange(0, h - patch, stride, dtype=np.int64)
    xs = np.arange(0, w - patch, stride, dtype=np.int64)
    yy, xx = np.meshgrid(ys, xs, indexing="ij")
    interior = np.stack([yy.ravel(), xx.ravel()], axis=1)
    last_col = np.stack([ys, np.full_like(ys, w - patch)], axis=1)
    last_row = np.stack([np.full_like(xs, h - patch), xs], axis=1)
    corner = np.array([[h - patch, w - patch]], dtype=np.int64)
    return np.concatenate([interior, last_col, last_row, corner]).astype(np.int32)


def batched_origins(h: int, w: int, patch: int, stride: int, batch: int) -> np.ndarray:
    """int32[n_batches, batch, 2]: origins chunked by `batch`; the last chunk is
    padded with copies of the final origin (full_samplers.py:375-377, 400-402)."""
    o = tile_origins(h, w, patch, stride)
    n = len(o)
    nb = -(-n // batch)
    pad = nb * batch - n
    if pad:
        o = np.concatenate([o, np.repeat(o[-1:], pad, axis=0)])
    return o.reshape(nb, batch, 2)


def progress_values(n_batches: int) -> list[float]:
    """progress = i / len(batches) (full_samplers.py:425-429); never reaches 1."""
    return [i / n_batches for i in range(n_batches)]


def gather_u8(slide: np.ndarray, origins: np.ndarray, patch: int) -> np.ndarray:
    """uint8[n, P, P, 3]: the patch views of full_samplers.py:353-369, stacked."""
    return np.stack([slide[y:y + patch, x:x + patch, :] for y, x in origins])


def features_nhwc(slide: np.ndarray, origins: np.ndarray, patch: int) -> np.ndarray:
    """float32[n, P, P, 3] = stack(u8).astype(f32) / 255 (a4, full_samplers.py:441-443)."""
    return gather_u8(slide, origins, patch).astype(np.float32) / 255


def features_nchw_predictor(slide: np.ndarray, origins: np.ndarray, patch: int) -> np.ndarray:
    """float32[n, 3, P, P] exactly as batch_predictor builds the model input (a5):
    np.stack(u8) / 255 in float64, cast to float32, NHWC -> NCHW
    (examples/predict_full_patched.py:67-71)."""
    f64 = gather_u8(slide, origins, patch) / 255
    return np.ascontiguousarray(f64.astype(np.float32).transpose(0, 3, 1, 2))


def coords_f32(origins: np.ndarray) -> np.ndarray:
    """float32[n, 2] (pos_y, pos_x) (full_samplers.py:444-451)."""
    return origins.astype(np.float32)


def accumulate_logits(h: int, w: int, n_cls: int, downscale: int, patch: int,
                      origins: np.ndarray, logits: np.ndarray) -> np.ndarray:
    """float32[h//d, w//d, n_cls] canvas (a8): sequential `+=` of each tile's logit
    vector over rows y//d:(y+P)//d and cols x//d:(x+P)//d, in list order, pad
    duplicates included (examples/predict_full_patched.py:41-54)."""
    d = downscale
    canvas = np.zeros([h // d, w // d, n_cls], dtype=np.float32)
    for (y, x), v in zip(origins.tolist(), logits):
        canvas[y // d:(y + patch) // d, x // d:(x + patch) // d, :] += v
    return canvas


def class_map(canvas: np.ndarray) -> np.ndarray:
    """int64[h//d, w//d] = argmax over classes (predict_full_patched.py:62)."""
    return np.argmax(canvas, axis=2)
