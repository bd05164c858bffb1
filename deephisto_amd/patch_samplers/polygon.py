"""Polygon geometry of the annotated-region samplers (host side, NumPy, float64).

The reference builds `shapely.Polygon`s and asks for `polygon.area`, `polygon.bounds`,
`polygon.is_valid` and `polygon.intersection(patch_square).area`
(patch_samplers/region_samplers.py:66-71, 115-137, 171-191).  shapely is a third-party
dependency that is absent here, so the four quantities are restated in closed form; nothing
in the reference pins them (its sampling is random and unseeded): **parity unpinned**, checked
instead against analytic cases and a brute-force rasterisation (tests/test_polygon.py).

Overlap with an axis-aligned rectangle R = [x0,x1] x [y0,y1] needs no clipping output: for a
counter-clockwise simple polygon P (convex or not)

    area(P n R) = sum over edges of  integral  F(x) dy   along the part of the edge with y in [y0, y1],
    F(x) = clamp(x, x0, x1) - x0,

because on every horizontal line the covered length inside R is the signed sum of F at the
boundary crossings (upward edges are right boundaries, downward edges left ones).  F is
piecewise linear, so each edge integral is (dy) * (G(xb) - G(xa)) / (xb - xa) with G the
antiderivative of F -- vectorised over edges AND over many candidate rectangles at once.
"""
from __future__ import annotations

import numpy as np


def signed_area(v: np.ndarray) -> float:
    """Shoelace area of the closed polygon v[n, 2] (x, y); > 0 for counter-clockwise."""
    x, y = v[:, 0], v[:, 1]
    return 0.5 * float(np.dot(x, np.roll(y, -1)) - np.dot(np.roll(x, -1), y))


def area(v: np.ndarray) -> float:
    return abs(signed_area(v))


def bounds(v: np.ndarray) -> tuple[float, float, float, float]:
    """(minx, miny, maxx, maxy) -- shapely's `polygon.bounds` order."""
    return float(v[:, 0].min()), float(v[:, 1].min()), float(v[:, 0].max()), float(v[:, 1].max())


def as_ccw(v: np.ndarray) -> np.ndarray:
    """Drop a repeated closing vertex and orient counter-clockwise."""
    v = np.asarray(v, dtype=np.float64)
    if len(v) > 1 and np.array_equal(v[0], v[-1]):
        v = v[:-1]
    return v if signed_area(v) >= 0 else v[::-1].copy()


def is_simple(v: np.ndarray) -> bool:
    """True when no two non-adjacent edges meet and no edge is degenerate (what shapely's
    `is_valid` asks of a ring, besides a non-zero area).  O(n^2), vectorised."""
    n = len(v)
    if n < 3:
        return False
    a, b = v, np.roll(v, -1, axis=0)
    if np.any(np.all(a == b, axis=1)):
        return False

    def cross(u, w):
        return u[..., 0] * w[..., 1] - u[..., 1] * w[..., 0]

    ai, bi = a[:, None, :], b[:, None, :]       # edge i along axis 0
    aj, bj = a[None, :, :], b[None, :, :]       # edge j along axis 1
    d1, d2 = cross(bi - ai, aj - ai), cross(bi - ai, bj - ai)
    d3, d4 = cross(bj - aj, ai - aj), cross(bj - aj, bi - aj)
    proper = (d1 * d2 < 0) & (d3 * d4 < 0)

    def on_seg(p, s0, s1, o):                   # collinear and inside the segment's box
        return (o == 0) & np.all((p >= np.minimum(s0, s1)) & (p <= np.maximum(s0, s1)), axis=-1)

    touch = on_seg(aj, ai, bi, d1) | on_seg(bj, ai, bi, d2) | on_seg(ai, aj, bj, d3) | on_seg(bi, aj, bj, d4)
    idx = np.arange(n)
    adjacent = (idx[:, None] == idx[None, :]) | ((idx[:, None] + 1) % n == idx[None, :]) | ((idx[None, :] + 1) % n == idx[:, None])
    return not bool(np.any((proper | touch) & ~adjacent))


def overlap_area_rect(v_ccw: np.ndarray, x0, y0, x1, y1) -> np.ndarray:
    """area(polygon n [x0,x1]x[y0,y1]) for a counter-clockwise simple polygon; the rectangle
    arguments broadcast (arrays of R candidates give float64[R])."""
    x0, y0, x1, y1 = (np.asarray(t, dtype=np.float64)[..., None] for t in (x0, y0, x1, y1))
    a, b = v_ccw, np.roll(v_ccw, -1, axis=0)
    ax, ay, bx, by = a[:, 0], a[:, 1], b[:, 0], b[:, 1]
    dy = by - ay
    nz = dy != 0
    safe = np.where(nz, dy, 1.0)
    t0, t1 = (y0 - ay) / safe, (y1 - ay) / safe                     # where the edge meets the two horizontal sides
    tlo = np.clip(np.minimum(t0, t1), 0.0, 1.0)
    thi = np.clip(np.maximum(t0, t1), 0.0, 1.0)
    live = nz & (thi > tlo)
    dx = bx - ax
    pax, pay = ax + tlo * dx, ay + tlo * dy
    pbx, pby = ax + thi * dx, ay + thi * dy

    def G(x):                                                       # antiderivative of F
        u = np.clip(x, x0, x1)
        return 0.5 * (u - x0) ** 2 + (x1 - x0) * np.maximum(x - x1, 0.0)

    ddx = pbx - pax
    steep = np.abs(ddx) < 1e-12
    mean_f = np.where(steep, np.clip(pax, x0, x1) - x0, (G(pbx) - G(pax)) / np.where(steep, 1.0, ddx))
    out = np.where(live, (pby - pay) * mean_f, 0.0).sum(axis=-1)
    return np.maximum(out, 0.0)


def overlap_area_square(v_ccw: np.ndarray, x, y, side) -> np.ndarray:
    """area(polygon n the patch square with top-left (x, y)); x, y broadcast."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    return overlap_area_rect(v_ccw, x, y, x + side, y + side)
