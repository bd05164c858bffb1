"""Closed-form synthetic whole-slide image (oracle side, NumPy).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

The reference has no synthetic data; real slides come from `.psi` files
through `psimage.PSImage.get_region_from_layer` (full_samplers.py:328-330)
as `uint8[h, w, 3]` HWC arrays.  For benchmarking we need slides that the
GPU can generate in HBM and the CPU oracle can reproduce tile by tile
without materialising 7.5 GB, so the pixel is a closed-form integer hash
(SURVEY.md section 8d).  The formula is frozen by tests/test_synth.py.

    pix(y, x, c, seed) = (((y*73856093) ^ (x*19349663) ^ (c*83492791)
                           ^ (seed*2654435761)) >> 7) & 0xFF      (uint32 wrap)
"""
import numpy as np

K_Y = np.uint32(73856093)
K_X = np.uint32(19349663)
K_C = np.uint32(83492791)
K_S = np.uint32(2654435761)


def synth_region(y0: int, x0: int, hh: int, ww: int, seed: int = 0) -> np.ndarray:
    """uint8[hh, ww, 3] block of the synthetic slide with origin (y0, x0)."""
    with np.errstate(over="ignore"):
        ys = (np.arange(y0, y0 + hh, dtype=np.uint64).astype(np.uint32) * K_Y)[:, None, None]
        xs = (np.arange(x0, x0 + ww, dtype=np.uint64).astype(np.uint32) * K_X)[None, :, None]
        cs = (np.arange(3, dtype=np.uint32) * K_C)[None, None, :]
        sd = np.uint32(seed & 0xFFFFFFFF) * K_S
        v = ys ^ xs ^ cs ^ sd
    return ((v >> np.uint32(7)) & np.uint32(0xFF)).astype(np.uint8)


def synth_slide(h: int, w: int, seed: int = 0) -> np.ndarray:
    """Whole uint8[h, w, 3] synthetic slide (use only for sizes that fit RAM)."""
    return synth_region(0, 0, h, w, seed)
