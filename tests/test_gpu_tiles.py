"""GPU parity, tile side (rows a1-a5, a8): HIP kernels vs the oracle and vs the
fixtures produced by the reference.  Everything here is integer/byte work or
exactly-rounded float32, so the bar is BIT-EXACT."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import synth, tiling

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def dev(built_lib):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def test_synth_slide_matches_oracle(dev):
    from deephisto_amd import tiles
    for (h, w, seed) in [(64, 48, 0), (333, 517, 3), (1024, 1024, 1), (5, 7, 9)]:
        got = tiles.synth_slide(h, w, seed, dev).cpu().numpy()
        np.testing.assert_array_equal(got, synth.synth_slide(h, w, seed))
    # a window of the full-size benchmark slide, without materialising it on the host
    big = tiles.synth_slide(50000, 50000, 2, dev)
    for (y, x) in [(0, 0), (49744, 49744), (31234, 777)]:
        np.testing.assert_array_equal(big[y:y + 256, x:x + 256].cpu().numpy(), synth.synth_region(y, x, 256, 256, 2))
    del big


def test_div255_all_bytes_bit_exact(dev):
    from deephisto_amd import tiles
    from deephisto_amd._lib import DH_LAYOUT_NHWC
    slide = torch.arange(256 * 4 * 3, dtype=torch.int32).remainder(256).to(torch.uint8).reshape(4, 256, 3).repeat(8, 1, 1)
    slide = slide.to(dev).contiguous()  # 32 x 256 x 3, every byte value present
    out = tiles.gather_tiles(slide, np.array([[0, 0]], np.int32), 32, DH_LAYOUT_NHWC, torch.float32)
    want = slide[:32, :32].cpu().numpy().astype(np.float32) / 255
    assert np.array_equal(out[0].cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("P", [256, 224, 100, 37])
def test_gather_bit_exact(dev, P):
    from deephisto_amd import tiles
    from deephisto_amd._lib import DH_LAYOUT_NCHW, DH_LAYOUT_NHWC
    h, w, seed = 700, 1013, 5          # odd width: rows are not 4-byte aligned
    host = synth.synth_slide(h, w, seed)
    slide = torch.from_numpy(host).to(dev)
    rng = np.random.default_rng(P)
    o = np.stack([rng.integers(0, h - P + 1, 9), rng.integers(0, w - P + 1, 9)], axis=1).astype(np.int32)
    o[0] = (0, 0); o[1] = (h - P, w - P)
    nhwc = tiling.features_nhwc(host, o, P)
    nchw = tiling.features_nchw_predictor(host, o, P)
    got = tiles.gather_tiles(slide, o, P, DH_LAYOUT_NHWC, torch.float32).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), nhwc.view(np.uint32))
    got = tiles.gather_tiles(slide, o, P, DH_LAYOUT_NCHW, torch.float32).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), nchw.view(np.uint32))
    for layout, ref in ((DH_LAYOUT_NHWC, nhwc), (DH_LAYOUT_NCHW, nchw)):
        got = tiles.gather_tiles(slide, o, P, layout, torch.bfloat16)
        want = torch.from_numpy(ref).to(torch.bfloat16)
        assert torch.equal(got.cpu().view(torch.int16), want.view(torch.int16))


def test_gather_rejects_out_of_bounds(dev):
    from deephisto_amd import _lib, tiles
    slide = torch.zeros((300, 300, 3), dtype=torch.uint8, device=dev)
    with pytest.raises(_lib.DeephistoHipError, match="outside slide"):
        tiles.gather_tiles(slide, np.array([[100, 100]], np.int32), 256)
    assert tiles.gather_tiles(slide, np.zeros((0, 2), np.int32), 256).shape == (0, 3, 256, 256)


def test_sampler_generator_torch_matches_reference_fixture(dev, golden_meta, golden_vectors):
    """The sampler's own iterator against hashes of what the REFERENCE's generator_torch
    produced on the same closed-form slide (tests/golden, oracle/make_golden.py)."""
    from deephisto_amd.patch_samplers.full_samplers import FullImageDenseSampler, SamplerExecutionMode
    for name, r in golden_meta["features"].items():
        host = synth.synth_slide(r["h"], r["w"], r["seed"])
        smp = FullImageDenseSampler(host, layer=1, patch_size=r["patch"], batch_size=r["batch"],
                                    mode=SamplerExecutionMode.INMEMORY_SINGLEPROC, stride=r["stride"], device=dev)
        n = 0
        for i, (f, c, prog) in enumerate(smp.generator_torch()):
            assert f.dtype == torch.float32 and tuple(f.shape) == (r["batch"], r["patch"], r["patch"], 3)
            assert sha(f.cpu().numpy()) == r["feature_sha256"][i], (name, i)
            np.testing.assert_array_equal(c.cpu().numpy(), golden_vectors[name + "_coords"][i])
            assert prog == r["progress"][i]
            n += 1
        assert n == len(r["progress"])
        # generator(): Patch records, lazy host views, same pixels as the reference's views
        for i, (patches, prog) in enumerate(smp.generator()):
            assert prog == r["progress"][i] and len(patches) == r["batch"]
            assert sha(np.stack([p.data for p in patches])) == r["u8_sha256"][i]
            assert patches[0].data.base is not None


def test_random_sampler_matches_reference_fixture(dev, golden_meta, golden_vectors):
    """FullImageRndSampler under the recorded NumPy seed against what the REFERENCE's sampler drew
    (origins, filled ratios, uint8 pixels; first generator_torch batch: raw 0..255 floats + coords)."""
    from deephisto_amd.patch_samplers.full_samplers import FullImageRndSampler, SamplerExecutionMode
    for name, r in golden_meta["random_sampler"].items():
        host = synth.synth_slide(r["h"], r["w"], r["seed"])
        mk = lambda: FullImageRndSampler(host, layer=1, patch_size=r["patch"], batch_size=r["batch"],
                                         mode=SamplerExecutionMode.INMEMORY_SINGLEPROC, dense_level=r["dense_level"],
                                         speedup=r["speedup"], device=dev)
        np.random.seed(r["np_seed"])
        smp = mk()
        assert (smp.dh, smp.dw) == (r["h"] // r["speedup"], r["w"] // r["speedup"])
        n = 0
        for i, (patches, filled) in enumerate(smp):
            got = np.array([(p.pos_y, p.pos_x) for p in patches], np.int32)
            np.testing.assert_array_equal(got, golden_vectors[name + "_origins"][i])
            assert filled == golden_vectors[name + "_ratios"][i]
            assert sha(np.stack([p.data for p in patches])) == r["u8_sha256"][i]
            n += 1
        assert n == r["n_batches"] and smp._filled_ratio[-1] >= 1.0
        # device iterator: same RNG stream -> same origins; features are the raw bytes as f32
        np.random.seed(r["np_seed"])
        smp = mk()
        for i, (f, c, filled) in enumerate(smp.generator_torch()):
            assert f.dtype == torch.float32 and tuple(f.shape) == (r["batch"], r["patch"], r["patch"], 3)
            assert c.dtype == torch.float32
            o = golden_vectors[name + "_origins"][i]
            np.testing.assert_array_equal(c.cpu().numpy(), o.astype(np.float32))
            want = tiling.gather_u8(host, o, r["patch"]).astype(np.float32)
            np.testing.assert_array_equal(f.cpu().numpy(), want)
            if i == 0:
                np.testing.assert_array_equal(f[:, :4, :4, :].cpu().numpy(), golden_vectors[name + "_torch_first_crop"])
                np.testing.assert_array_equal(c.cpu().numpy(), golden_vectors[name + "_torch_first_coords"])
            assert filled == golden_vectors[name + "_ratios"][i]
        assert i + 1 == r["n_batches"]


@pytest.mark.parametrize("case", ["p1000x1300_256_256_16_d16", "p600x900_224_112_16_d16", "p700x500_100_60_8_d7"])
def test_accumulate_and_argmax_bit_exact(dev, golden_meta, golden_vectors, case):
    from deephisto_amd import tiles
    r = golden_meta["predict"][case]
    o = tiling.batched_origins(r["h"], r["w"], r["patch"], r["stride"], r["batch"]).reshape(-1, 2)
    for kind in ("toy", "torch"):
        logits = golden_vectors[f"{case}_{kind}_logits"]
        want = tiling.accumulate_logits(r["h"], r["w"], 5, r["downscale"], r["patch"], o, logits)
        canvas, cmap = tiles.accumulate_logits(torch.from_numpy(logits).to(dev), o, r["patch"], r["downscale"],
                                               r["h"], r["w"])
        assert np.array_equal(canvas.cpu().numpy().view(np.uint32), want.view(np.uint32))
        np.testing.assert_array_equal(cmap.cpu().numpy(), golden_vectors[f"{case}_{kind}_map"].astype(np.int64))
    # split into two calls: the canvas accumulates across calls in order
    logits = torch.from_numpy(golden_vectors[f"{case}_toy_logits"]).to(dev)
    k = len(o) // 3
    c1, _ = tiles.accumulate_logits(logits[:k].contiguous(), o[:k], r["patch"], r["downscale"], r["h"], r["w"])
    c2, m2 = tiles.accumulate_logits(logits[k:].contiguous(), o[k:], r["patch"], r["downscale"], r["h"], r["w"], canvas=c1)
    want = tiling.accumulate_logits(r["h"], r["w"], 5, r["downscale"], r["patch"], o, golden_vectors[f"{case}_toy_logits"])
    assert np.array_equal(c2.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_accumulate_random_origins_and_ties(dev):
    """Arbitrary (random-sampler style) origins, heavy overlap, ties and NaN in argmax."""
    from deephisto_amd import tiles
    rng = np.random.default_rng(0)
    h, w, P, d, n = 900, 1100, 224, 16, 300
    o = np.stack([rng.integers(0, h - P + 1, n), rng.integers(0, w - P + 1, n)], axis=1).astype(np.int32)
    logits = rng.standard_normal((n, 5)).astype(np.float32)
    logits[::7] = logits[::7].round()  # provoke exact ties
    want = tiling.accumulate_logits(h, w, 5, d, P, o, logits)
    canvas, cmap = tiles.accumulate_logits(torch.from_numpy(logits).to(dev), o, P, d, h, w)
    assert np.array_equal(canvas.cpu().numpy().view(np.uint32), want.view(np.uint32))
    np.testing.assert_array_equal(cmap.cpu().numpy(), np.argmax(want, axis=2))
    t = torch.tensor([[1.0, 1.0, 0.0], [0.0, float("nan"), 5.0], [-1.0, -1.0, -1.0]], device=dev)
    m = torch.empty(3, dtype=torch.int64, device=dev)
    from deephisto_amd._lib import check, lib
    check(lib().dh_argmax_map(t.data_ptr(), 3, 3, m.data_ptr(), None), "argmax")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(m.cpu().numpy(), np.argmax(t.cpu().numpy(), axis=1))


def test_full_size_grid_and_tiles_properties(dev):
    """BASELINE size (50000^2, 256/256/64): grid hash vs the reference fixture is covered on
    the CPU; here: every tile of the full slide gathered on the GPU has the checksum the
    closed-form oracle predicts for a sample of origins, and coverage is complete."""
    from deephisto_amd import tiles
    from deephisto_amd._lib import DH_LAYOUT_NCHW
    o, n_unique = tiles.tile_grid(50000, 50000, 256, 256, 64)
    assert n_unique == 38416 and len(o) == 601 * 64
    cover = np.zeros((50000 // 16, 50000 // 16), np.int32)
    for y, x in o[:n_unique]:
        cover[y // 16:(y + 256) // 16, x // 16:(x + 256) // 16] += 1
    assert cover.min() >= 1
    slide = tiles.synth_slide(50000, 50000, 0, dev)
    idx = np.array([0, 195, 38024, 38219, 38220, 38415, 20000, 38415])
    got = tiles.gather_tiles(slide, o[idx], 256, DH_LAYOUT_NCHW, torch.float32).cpu().numpy()
    for j, i in enumerate(idx):
        y, x = o[i]
        want = (synth.synth_region(int(y), int(x), 256, 256, 0).astype(np.float32) / 255).transpose(2, 0, 1)
        assert np.array_equal(got[j].view(np.uint32), np.ascontiguousarray(want).view(np.uint32))


def test_sample_full_dense_example(dev, capsys):
    """The caller of examples/sample_full_dense.py on the BASELINE configs[0] geometry (4096^2, 256 / 256 / 64): 4 batches of
    [64, 256, 256, 3] float32 features and [64, 2] coords, progress 0, 0.25, 0.5, 0.75, then the items/s line."""
    from deephisto_amd.examples.sample_full_dense import main
    n, _ = main(["--side", "4096"])
    out = capsys.readouterr().out.strip().splitlines()
    assert n == 256
    shapes = [l for l in out if l.startswith("torch.Size")]
    assert len(shapes) == 4 and shapes[0].startswith("torch.Size([64, 256, 256, 3]) torch.Size([64, 2]) 0.0")
    assert shapes[3].endswith("0.75") and out[-1].endswith("items/s")
