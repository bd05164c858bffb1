"""Host geometry and random stream of the annotated-region samplers.

Parity unpinned: shapely is absent and the reference holds no fixture for this path, so the
restated geometry is checked against analytic cases and a brute-force rasterisation, and the
samplers against the properties the reference's code guarantees (overlap threshold, weights,
chunking of the random stream, output records)."""
import json

import numpy as np
import pytest

from deephisto_amd.patch_samplers import polygon as pg


def _raster_area(v, x0, y0, x1, y1, n=600):
    xs = np.linspace(x0, x1, n, endpoint=False) + (x1 - x0) / n / 2
    ys = np.linspace(y0, y1, n, endpoint=False) + (y1 - y0) / n / 2
    X, Y = np.meshgrid(xs, ys)
    inside = np.zeros_like(X, bool)
    for (ax, ay), (bx, by) in zip(v, np.roll(v, -1, 0)):
        cond = (ay > Y) != (by > Y)
        xint = ax + (Y - ay) * (bx - ax) / np.where(by == ay, 1, by - ay)
        inside ^= cond & (X < xint)
    return inside.mean() * (x1 - x0) * (y1 - y0)


def _star(seed, n=19, cx=500.0, cy=400.0, r0=120.0, r1=380.0):
    rng = np.random.default_rng(seed)
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    r = rng.uniform(r0, r1, n)
    return np.stack([cx + r * np.cos(ang), cy + r * np.sin(ang)], 1)


def test_area_bounds_orientation_and_validity():
    sq = np.array([[0, 0], [0, 10], [10, 10], [10, 0]], float)        # clockwise in (x, y)
    assert pg.signed_area(sq) == -100.0 and pg.area(sq) == 100.0
    assert pg.signed_area(pg.as_ccw(sq)) == 100.0
    assert pg.bounds(sq) == (0.0, 0.0, 10.0, 10.0)
    assert pg.is_simple(sq)
    assert not pg.is_simple(np.array([[0, 0], [10, 10], [10, 0], [0, 10]], float))   # bow tie
    assert not pg.is_simple(np.array([[0, 0], [5, 5]], float))
    closed = np.vstack([sq, sq[:1]])
    assert len(pg.as_ccw(closed)) == 4


def test_overlap_analytic_cases():
    sq = pg.as_ccw(np.array([[0, 0], [10, 0], [10, 10], [0, 10]], float))
    assert float(pg.overlap_area_rect(sq, 5, 5, 20, 20)) == 25.0
    assert float(pg.overlap_area_rect(sq, -5, -5, 0, 0)) == 0.0
    assert float(pg.overlap_area_rect(sq, -1, -1, 11, 11)) == 100.0
    np.testing.assert_allclose(pg.overlap_area_square(sq, np.array([-2.0, 3, 9]), np.array([-2.0, 3, 9]), 4), [4, 16, 1])
    L = pg.as_ccw(np.array([[0, 0], [10, 0], [10, 4], [4, 4], [4, 10], [0, 10]], float))   # concave
    assert pg.area(L) == 64.0
    assert float(pg.overlap_area_rect(L, 2, 2, 8, 8)) == 20.0
    tri = pg.as_ccw(np.array([[0, 0], [8, 0], [0, 8]], float))
    np.testing.assert_allclose(float(pg.overlap_area_rect(tri, 0, 0, 4, 4)), 16.0)
    np.testing.assert_allclose(float(pg.overlap_area_rect(tri, 2, 2, 6, 6)), 8.0)      # the half of the square below x+y=8


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_overlap_matches_rasterisation(seed):
    star = pg.as_ccw(_star(seed))
    assert pg.is_simple(star)
    rng = np.random.default_rng(100 + seed)
    for _ in range(6):
        x, y = rng.uniform(100, 800), rng.uniform(50, 700)
        s = rng.uniform(40, 300)
        want = _raster_area(star, x, y, x + s, y + s)
        got = float(pg.overlap_area_square(star, x, y, s))
        assert abs(got - want) <= 2e-3 * s * s + 1.0
    np.testing.assert_allclose(float(pg.overlap_area_rect(star, 0, 0, 2000, 2000)), pg.area(star), rtol=1e-12)


def _two_image_annotations():
    a0 = [{"class": "TUM", "vertices": _star(3, cx=600, cy=600, r0=250, r1=500).tolist()},
          {"class": "BG", "vertices": [[100, 1300], [900, 1300], [900, 1900], [100, 1900]]},
          {"class": "BG", "vertices": [[0, 0], [50, 50], [50, 0], [0, 50]]},            # bow tie: repaired (too small to sample)
          {"class": "SKIP", "vertices": [[0, 0], [300, 0], [300, 300], [0, 300]]}]
    a1 = [{"class": "TUM", "vertices": [[200, 200], [1400, 260], [1300, 1500], [260, 1400]]},
          {"class": "LP", "vertices": _star(4, cx=1500, cy=600, r0=200, r1=400).tolist()}]
    return a0, a1


def test_region_annotation_and_random_origins():
    from deephisto_amd.patch_samplers.region_samplers import RegionAnnotation
    v = _star(5, cx=900, cy=900, r0=400, r1=800)
    reg = RegionAnnotation("img", 0, "TUM", v, layer=1, layer_size=(2000, 2000))
    assert reg.area == pytest.approx(pg.area(v)) and reg.class_ == "TUM"
    with pytest.raises(RuntimeError, match="dtype"):
        RegionAnnotation("img", 0, "TUM", v.astype(np.float32), 1, (2000, 2000))
    with pytest.raises(RuntimeError, match="shape"):
        RegionAnnotation("img", 0, "TUM", v.ravel(), 1, (2000, 2000))
    with pytest.raises(RuntimeError, match="polygon"):      # a ring without area cannot be repaired either
        RegionAnnotation("img", 0, "TUM", np.array([[0, 0], [10, 10], [20, 20]], float), 1, (2000, 2000))
    np.random.seed(7)
    c1 = reg._extract_patch_coords_rnd(224, 12, 0.75)
    np.random.seed(7)
    c2 = reg._extract_patch_coords_rnd(224, 12, 0.75)
    assert c1 == c2 and len(c1) == 12
    for y, x in c1:
        assert float(pg.overlap_area_square(reg.polygon, x, y, 224)) > 0.75 * 224 * 224
        assert 0 <= y and 0 <= x
    small = RegionAnnotation("img", 1, "TUM", np.array([[0, 0], [100, 0], [100, 100], [0, 100]], float), 1, (2000, 2000))
    with pytest.raises(RuntimeError, match="too small"):
        small._extract_patch_coords_rnd(224, 1)
    # layer 2: vertices are halved
    half = RegionAnnotation("img", 2, "TUM", v, layer=2, layer_size=(1000, 1000))
    assert half.area == pytest.approx(reg.area / 4)
    dense = reg._extract_patch_coords_dense(224, 112, 0.75)
    assert dense and dense == sorted(dense)
    assert all(float(pg.overlap_area_square(reg.polygon, x, y, 224)) > 0.75 * 224 * 224 for y, x in dense)
    x0, y0, x1, y1 = (round(t) for t in reg.bounds)
    assert all((y - y0) % 112 == 0 and (x - x0) % 112 == 0 for y, x in dense)


def test_sampler_weights_stream_and_records(tmp_path):
    from deephisto_amd.patch_samplers.region_samplers import AnnoRegionDenseSampler, AnnoRegionRndSampler
    a0, a1 = _two_image_annotations()
    p0 = tmp_path / "a0.json"
    p0.write_text(json.dumps(a0))
    img0 = (np.arange(2000 * 2000 * 3, dtype=np.uint32) % 251).astype(np.uint8).reshape(2000, 2000, 3)
    img1 = (np.arange(1700 * 2100 * 3, dtype=np.uint32) % 241).astype(np.uint8).reshape(1700, 2100, 3)
    smp = AnnoRegionRndSampler([(img0, p0), (img1, a1)], layer=1, patch_size=128, classes=["TUM", "BG", "LP"],
                               patches_from_one_region=4, region_area_influence=0.5)
    assert smp.classes == ["BG", "LP", "TUM"]
    assert {c: len(r) for c, r in smp.regions.items()} == {"TUM": 2, "BG": 2, "LP": 1}      # SKIP dropped; the bow tie is repaired
    assert sorted(r.area for r in smp.regions["BG"])[0] == pytest.approx(625.0)               # ... to its 25 x 50 / 2 lobe
    assert [sorted(d) for d in smp.regions_per_image] == [["BG", "TUM"], ["LP", "TUM"]]
    for w in list(smp._reg_w_all.values()) + [smp._img_w_all] + list(smp._img_w.values()):
        assert abs(float(np.sum(w)) - 1.0) < 1e-12
    # area influence: 0 -> uniform; +1 -> proportional; -1 -> inverse proportional
    areas = [100.0, 300.0]
    np.testing.assert_allclose(smp._calc_area_weights(areas, 0), [0.5, 0.5])
    np.testing.assert_allclose(smp._calc_area_weights(areas, 1), [0.25, 0.75])
    np.testing.assert_allclose(smp._calc_area_weights(areas, -1), [0.75, 0.25])
    np.testing.assert_allclose(smp._calc_area_weights(areas, 0.5), [0.375, 0.625])
    assert len(smp) == int(sum(r.area for rs in smp.regions.values() for r in rs) / 128 ** 2)
    assert smp._split_chunks(5, 2) == [2, 2, 1]
    # the random stream is a function of the global NumPy seed; records respect class, image and overlap
    np.random.seed(11)
    r1 = smp._records(10)
    np.random.seed(11)
    r2 = smp._records(10)
    assert r1 == r2 and len(r1) == 10
    by_region = {}
    for j, y, x, c in r1:
        regs = smp.regions_per_image[j][smp.classes[c]]
        assert any(float(pg.overlap_area_square(r.polygon, x, y, 128)) > 0.75 * 128 * 128 for r in regs)
    # structs_generator: lists of (Patch, class index), pixels = views of the host image
    np.random.seed(3)
    batches = list(smp.structs_generator(batch_size=6, n_batches=3, batches_per_worker=2))
    assert [len(b) for b in batches] == [6, 6, 6]
    for p, c in batches[0]:
        assert p.patch_size == 128 and p.data.shape == (128, 128, 3) and 0 <= c < 3
    np.random.seed(3)
    recs = smp._records(12)            # first chunk = 2 batches of 6 from ONE run of the stream
    imgs = [img0, img1]
    for (p, c), (j, y, x, cc) in zip(batches[0] + batches[1], recs):
        assert (p.pos_y, p.pos_x, c) == (y, x, cc)
        np.testing.assert_array_equal(p.data, imgs[j][y:y + 128, x:x + 128])
    # one_image_for_batch: every record of a run comes from one image
    one = AnnoRegionRndSampler([(img0, a0), (img1, a1)], layer=1, patch_size=128, classes=["TUM", "BG", "LP"],
                               one_image_for_batch=True)
    np.random.seed(5)
    assert len({j for j, *_ in one._records(9)}) == 1
    # dense sampler: class by class, region by region, grid order
    dense = AnnoRegionDenseSampler([(img0, a0), (img1, a1)], layer=1, patch_size=128, stride=128, classes=["BG", "LP"])
    got = list(dense.structs_generator())
    assert got and [c for _, c in got] == sorted(c for _, c in got)
    bg = [(p.pos_y, p.pos_x) for p, c in got if c == 0]
    assert bg == [(y, x) for y in range(1300, 1900 - 128, 128) for x in range(100, 900 - 128, 128)]



def test_self_crossing_rings_are_repaired_like_buffer0():
    """region_samplers.py:68-71: an invalid polygon is replaced by `polygon.buffer(0)`.  The restatement (polygon.repair) nodes
    the ring at its crossings / touches and keeps the lobes wound like the ring (the turn at its highest vertex): analytic
    cases, then a RegionAnnotation on a crossed ring samples only inside the kept lobe.  Parity unpinned (no shapely)."""
    from deephisto_amd.patch_samplers.region_samplers import RegionAnnotation
    # bow tie: the lobe with the ring's highest vertex survives, the mirror lobe (opposite winding) does not
    r = pg.repair(np.array([[0, 0], [10, 10], [10, 0], [0, 10]], float))
    assert len(r) == 1 and pg.area(r[0]) == pytest.approx(25.0)
    assert sorted(map(tuple, r[0].tolist())) == [(5.0, 5.0), (10.0, 0.0), (10.0, 10.0)]
    # a ring that only TOUCHES itself at a vertex: two lobes of one orientation, both kept
    r = pg.repair(np.array([[0, 0], [0, 2], [1, 1], [2, 2], [2, 0], [1, 1]], float))
    assert len(r) == 2 and sum(pg.area(x) for x in r) == pytest.approx(2.0)
    # a hand-drawn glitch: the last stroke doubles back over the outline (T-junction + tiny flag)
    g = np.array([[0, 0], [100, 0], [100, 100], [2, 100], [0, 98], [3, 103], [0, 100]], float)
    r = pg.repair(g)
    assert sum(pg.area(x) for x in r) == pytest.approx(9998.0 + 3.0) and all(pg.is_simple(x) for x in r)
    # simple rings pass through unchanged; repeated consecutive points are not an invalidity
    sq = np.array([[0, 0], [4, 0], [4, 0], [4, 3], [0, 3], [0, 0]], float)
    assert pg.is_simple(pg.as_ccw(pg.drop_repeats(sq))) and pg.area(pg.repair(sq)[0]) == pytest.approx(12.0)
    # overlap with rectangles is additive over the kept rings: against a rasterisation of "inside a kept lobe"
    big = np.array([[100, 100], [900, 900], [900, 100], [100, 900]], float)      # crossed: right lobe kept
    rings = pg.repair(big)
    assert len(rings) == 1 and pg.area(rings[0]) == pytest.approx(0.5 * 800 * 400)
    ys, xs = np.mgrid[0:1000, 0:1000] + 0.5
    inside = (xs >= 500) & (np.abs(ys - 500) <= xs - 500) & (xs <= 900)          # the right triangle
    for (x, y, sd) in [(600, 400, 200), (450, 450, 100), (100, 100, 300), (700, 650, 150)]:
        want = inside[y:y + sd, x:x + sd].sum()
        assert float(pg.overlap_area_square(rings, x, y, sd)) == pytest.approx(want, abs=sd * 1.5)
    reg = RegionAnnotation("img", 0, "TUM", big, layer=1, layer_size=(1000, 1000))
    assert reg.area == pytest.approx(160000.0) and reg.bounds == (500.0, 100.0, 900.0, 900.0)
    np.random.seed(3)
    for y, x in reg._extract_patch_coords_rnd(96, 10, 0.75):
        assert inside[y:y + 96, x:x + 96].mean() > 0.7
