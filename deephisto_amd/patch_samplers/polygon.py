"""Polygon geometry of the annotated-region samplers (host side, NumPy, float64).

The reference builds `shapely.Polygon`s and asks for `polygon.area`, `polygon.bounds`,
`polygon.is_valid` and `polygon.intersection(patch_square).area`
(patch_samplers/region_samplers.py:66-71, 115-137, 171-191).  shapely is a third-party
dependency that is absent here, so the four quantities are restated in closed form; nothing
in the reference pins them (its sampling is random and unseeded): **parity unpinned**, checked
instead against analytic cases and a brute-force rasterisation (tests/test_polygon.py).  `repair`
restates the `polygon.buffer(0)` the reference applies to invalid rings (:69-71).

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


def drop_repeats(v: np.ndarray) -> np.ndarray:
    """Remove consecutive repeated vertices (shapely accepts them: `is_valid` looks at the geometry, not the point list)."""
    v = np.asarray(v, dtype=np.float64)
    if len(v) > 1 and np.array_equal(v[0], v[-1]):
        v = v[:-1]
    keep = np.any(v != np.roll(v, 1, axis=0), axis=1)
    return v[keep] if keep.any() else v[:1]


def _proper_crossings(v: np.ndarray):
    """(i, j, t_i, t_j, point) of every proper crossing between non-adjacent edges i < j of the ring."""
    n = len(v)
    a, b = v, np.roll(v, -1, axis=0)
    out = []
    for i in range(n):
        d = b[i] - a[i]
        for j in range(i + 1, n):
            if j == i + 1 or (i == 0 and j == n - 1):
                continue
            e = b[j] - a[j]
            den = d[0] * e[1] - d[1] * e[0]
            if den == 0.0:
                continue
            w = a[j] - a[i]
            t = (w[0] * e[1] - w[1] * e[0]) / den
            u = (w[0] * d[1] - w[1] * d[0]) / den
            if 0.0 < t < 1.0 and 0.0 < u < 1.0:
                out.append((i, j, t, u, a[i] + t * d))
    return out


def split_loops(v: np.ndarray) -> list[np.ndarray]:
    """Cut a ring at its proper self-crossings and at vertices it visits twice into closed loops that no longer cross or
    touch themselves: walk the ring with the crossing points inserted; whenever a point comes up the second time, the
    stretch since its first visit is a loop."""
    v = drop_repeats(v)
    cross = _proper_crossings(v)
    n = len(v)
    per_edge = [[] for _ in range(n)]                  # (t, point)
    for i, j, t, u, pnt in cross:
        key = (float(pnt[0]), float(pnt[1]))
        per_edge[i].append((t, key))
        per_edge[j].append((u, key))
    a, b = v, np.roll(v, -1, axis=0)
    for i in range(n):                                 # T-junctions: a vertex lying inside another edge becomes a node of that edge
        d = b[i] - a[i]
        dd = float(d @ d)
        for k in range(n):
            if k == i or k == (i + 1) % n:
                continue
            w = v[k] - a[i]
            if d[0] * w[1] - d[1] * w[0] == 0.0:
                t = float(w @ d) / dd
                if 0.0 < t < 1.0:
                    per_edge[i].append((t, (float(v[k, 0]), float(v[k, 1]))))
    seq = []                                           # points as (x, y) tuples: equal coordinates = the same node
    for i in range(n):
        seq.append((float(v[i, 0]), float(v[i, 1])))
        seq.extend(key for _, key in sorted(per_edge[i]))
    loops, path, pos = [], [], {}
    for node in seq:
        if node in pos:
            k = pos[node]
            loop = path[k:]
            for q in loop[1:]:
                pos.pop(q, None)
            del path[k + 1:]
            loops.append(np.array(loop, dtype=np.float64))
        else:
            pos[node] = len(path)
            path.append(node)
    loops.append(np.array(path, dtype=np.float64))
    return [drop_repeats(l) for l in loops if len(l) >= 3]


def repair(v: np.ndarray) -> list[np.ndarray]:
    """Rings of the region a self-crossing ring stands for, each simple and counter-clockwise -- a restatement of what
    shapely's `polygon.buffer(0)` (GEOS zero-width buffer, region_samplers.py:69-71) keeps: the ring is noded at its crossings
    and only the lobes wound like the ring itself survive, "like the ring itself" being the turn at its highest vertex (the
    test GEOS' orientation routine applies); lobes wound the other way have negative depth and are dropped.  Rings that only
    touch themselves come out as several lobes of one orientation (all kept).  **Parity unpinned** (no shapely here): exact for
    lobes of winding +-1, which is what a stray crossing in a hand-drawn annotation produces."""
    v = drop_repeats(v)
    loops = [l for l in split_loops(v) if len(l) >= 3 and signed_area(l) != 0.0]
    if not loops:
        return []
    top = v[np.lexsort((-v[:, 0], -v[:, 1]))[0]]                     # highest vertex (first by y, then by x)
    ref = next((l for l in loops if np.any(np.all(l == top, axis=1))), loops[0])
    sign = np.sign(signed_area(ref))
    return [as_ccw(l) for l in loops if np.sign(signed_area(l)) == sign and is_simple(as_ccw(l))]


def overlap_area_rect(v_ccw, x0, y0, x1, y1) -> np.ndarray:
    """area(polygon n [x0,x1]x[y0,y1]) for a counter-clockwise simple polygon -- or a list of such rings with disjoint
    interiors (a repaired region: the edge integral is additive); the rectangle arguments broadcast (arrays of R candidates
    give float64[R])."""
    if isinstance(v_ccw, (list, tuple)):
        return sum(overlap_area_rect(r, x0, y0, x1, y1) for r in v_ccw)
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


def overlap_area_square(v_ccw, x, y, side) -> np.ndarray:
    """area(polygon n the patch square with top-left (x, y)); x, y broadcast."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    return overlap_area_rect(v_ccw, x, y, x + side, y + side)
