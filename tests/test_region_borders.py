"""Annotated regions that touch or leave the image border.

The reference's origin bounds (region_samplers.py:112-118, 160-166) keep patches of such regions that hang over
the border -- `randint(x0, min(max(x0 + 1, x1 - ps), w))` only caps the origin at w, not w - ps, and a polygon
that starts at a negative coordinate yields negative origins.  This build keeps those bounds (same random
stream) and defines the pixels outside the image as 0, on the host path here and in the device gather
(tests/test_gpu_train_loop.py::test_border_regions_zero_filled_on_device)."""
import numpy as np

from deephisto_amd.patch_samplers.region_samplers import AnnoRegionDenseSampler, AnnoRegionRndSampler
from oracle import synth


def _expected(img, y, x, ps):
    out = np.zeros((ps, ps, 3), np.uint8)
    h, w = img.shape[:2]
    ya, yb, xa, xb = max(y, 0), min(y + ps, h), max(x, 0), min(x + ps, w)
    if yb > ya and xb > xa:
        out[ya - y:yb - y, xa - x:xb - x] = img[ya:yb, xa:xb]
    return out


def test_edge_hugging_region_gives_overhanging_origins_and_zero_padding():
    img = synth.synth_slide(700, 1000, 5)
    # 210 x 600 region touching the right edge of the 1000-wide layer, ps = 256: x is always 790 (x + ps = 1046 > w)
    anno = [{"class": "TUM", "vertices": [[790, 50], [1000, 50], [1000, 650], [790, 650]]}]
    smp = AnnoRegionRndSampler([(img, anno)], layer=1, patch_size=256, region_intersection=0.5, device="cpu")
    np.random.seed(3)
    recs = smp._records(8)
    assert all(x == 790 for _, _, x, _ in recs)
    np.random.seed(3)
    for (p, c), (j, y, x, cls) in zip(smp._gen_single_proc(8), recs):
        assert p.data.shape == (256, 256, 3) and (p.pos_y, p.pos_x) == (y, x) and c == cls
        np.testing.assert_array_equal(p.data, _expected(img, y, x, 256))
        assert not p.data[:, 1000 - x:].any() and p.data[:, :1000 - x].any()


def test_negative_coordinate_polygon_dense_and_random():
    img = synth.synth_slide(600, 600, 6)
    anno = [{"class": "BG", "vertices": [[-50, -30], [400, -30], [400, 380], [-50, 380]]}]
    dense = AnnoRegionDenseSampler([(img, anno)], layer=1, patch_size=128, stride=64, region_intersection=0.75, device="cpu")
    got = [(p.pos_y, p.pos_x, p.data) for p, _ in dense.structs_generator()]
    assert (-30, -50) in [(y, x) for y, x, _ in got]          # the reference's grid starts at the rounded bounds
    for y, x, data in got:
        np.testing.assert_array_equal(data, _expected(img, y, x, 128))
    rnd = AnnoRegionRndSampler([(img, anno)], layer=1, patch_size=128, device="cpu")
    np.random.seed(11)
    recs = rnd._records(32)
    assert any(y < 0 or x < 0 for _, y, x, _ in recs)
    np.random.seed(11)
    for (p, _), (_, y, x, _) in zip(rnd._gen_single_proc(32), recs):
        np.testing.assert_array_equal(p.data, _expected(img, y, x, 128))
