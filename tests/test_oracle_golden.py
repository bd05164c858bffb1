"""The oracle (CPU restatement) against fixtures produced by the reference itself
(oracle/make_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest

from oracle import synth, tiling


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_tile_grid_matches_reference(golden_meta, golden_grids):
    for name, g in golden_meta["grids"].items():
        b = tiling.batched_origins(g["h"], g["w"], g["patch"], g["stride"], g["batch"])
        flat = b.reshape(-1, 2)
        assert b.shape[0] == g["n_batches"], name
        assert len(flat) == g["n_padded"], name
        assert len(tiling.tile_origins(g["h"], g["w"], g["patch"], g["stride"])) == g["n_unique"], name
        assert flat.dtype == np.int32
        assert sha(flat) == g["sha256_int32_yx_padded"], name
        if name in golden_grids:
            np.testing.assert_array_equal(flat, golden_grids[name])
        else:
            np.testing.assert_array_equal(flat[:64], golden_grids[name + "_head"])
            np.testing.assert_array_equal(flat[-64:], golden_grids[name + "_tail"])


def test_known_grid_facts():
    # SURVEY section 8a probes
    o = tiling.tile_origins(4096, 4096, 256, 256)
    assert len(o) == 256 and tuple(o[14]) == (0, 3584) and tuple(o[15]) == (256, 0)
    assert tuple(o[225]) == (0, 3840) and tuple(o[240]) == (3840, 0) and tuple(o[255]) == (3840, 3840)
    assert len(tiling.tile_origins(50000, 50000, 256, 256)) == 38416
    assert tiling.batched_origins(50000, 50000, 256, 256, 64).shape == (601, 64, 2)


def test_features_coords_progress_match_reference(golden_meta, golden_vectors):
    for name, r in golden_meta["features"].items():
        slide = synth.synth_slide(r["h"], r["w"], r["seed"])
        b = tiling.batched_origins(r["h"], r["w"], r["patch"], r["stride"], r["batch"])
        assert tiling.progress_values(len(b)) == r["progress"]
        for i, o in enumerate(b):
            assert sha(tiling.gather_u8(slide, o, r["patch"])) == r["u8_sha256"][i]
            f = tiling.features_nhwc(slide, o, r["patch"])
            assert f.dtype == np.float32
            assert sha(f) == r["feature_sha256"][i], (name, i)
            if i == 0:
                np.testing.assert_array_equal(f[:, :8, :8, :], golden_vectors[name + "_first_crop"])
            np.testing.assert_array_equal(tiling.coords_f32(o), golden_vectors[name + "_coords"][i])


def test_f64_and_f32_normalisation_agree():
    # batch_predictor divides in float64 then casts (predict_full_patched.py:67-70);
    # generator_torch divides in float32 (full_samplers.py:441-443): same bits.
    k = np.arange(256, dtype=np.uint8)
    assert np.array_equal((k / 255).astype(np.float32), k.astype(np.float32) / 255)


def test_accumulate_argmax_match_reference(golden_meta, golden_vectors):
    for name, r in golden_meta["predict"].items():
        o = tiling.batched_origins(r["h"], r["w"], r["patch"], r["stride"], r["batch"]).reshape(-1, 2)
        for kind in ("toy", "torch"):
            logits = golden_vectors[f"{name}_{kind}_logits"]
            canvas = tiling.accumulate_logits(r["h"], r["w"], 5, r["downscale"], r["patch"], o, logits)
            cmap = tiling.class_map(canvas)
            assert cmap.dtype == np.int64 and list(cmap.shape) == r["map_shape"]
            np.testing.assert_array_equal(cmap, golden_vectors[f"{name}_{kind}_map"].astype(np.int64))


def test_predictor_input_matches_reference_logits(golden_meta, golden_vectors):
    """a5: oracle NCHW features through the same toy torch model give the logits the
    reference's batch_predictor produced (same machine class -> tight tolerance)."""
    import torch
    from oracle.make_golden import toy_torch_model

    model = toy_torch_model(7)
    for name, r in golden_meta["predict"].items():
        slide = synth.synth_slide(r["h"], r["w"], r["seed"])
        o = tiling.batched_origins(r["h"], r["w"], r["patch"], r["stride"], r["batch"]).reshape(-1, 2)
        x = torch.from_numpy(tiling.features_nchw_predictor(slide, o, r["patch"]))
        with torch.no_grad():
            got = model(x).numpy()
        np.testing.assert_allclose(got, golden_vectors[f"{name}_torch_logits"], rtol=0, atol=2e-6)


def test_synth_formula_frozen():
    a = synth.synth_region(0, 0, 2, 3, seed=0)
    want = np.empty((2, 3, 3), np.uint8)
    for y in range(2):
        for x in range(3):
            for c in range(3):
                v = ((y * 73856093) ^ (x * 19349663) ^ (c * 83492791) ^ 0) & 0xFFFFFFFF
                want[y, x, c] = (v >> 7) & 0xFF
    np.testing.assert_array_equal(a, want)
    big = synth.synth_region(49000, 49990, 3, 5, seed=2)
    y, x, c = 49001, 49993, 2
    v = (((y * 73856093) & 0xFFFFFFFF) ^ ((x * 19349663) & 0xFFFFFFFF) ^ ((c * 83492791) & 0xFFFFFFFF)
         ^ ((2 * 2654435761) & 0xFFFFFFFF))
    assert big[1, 3, 2] == (v >> 7) & 0xFF
    assert sha(synth.synth_slide(64, 48, 1)) == sha(synth.synth_region(0, 0, 64, 48, 1))


def test_random_sampler_oracle_matches_reference(golden_meta, golden_vectors):
    """FullImageRndSampler under a fixed NumPy seed: origins, filled ratios, pixels."""
    from oracle import random_sampler
    for name, r in golden_meta["random_sampler"].items():
        slide = synth.synth_slide(r["h"], r["w"], r["seed"])
        np.random.seed(r["np_seed"])
        got = list(random_sampler.random_batches(r["h"], r["w"], r["patch"], r["batch"], r["dense_level"], r["speedup"]))
        assert len(got) == r["n_batches"]
        np.testing.assert_array_equal(np.stack([o for o, _ in got]), golden_vectors[name + "_origins"])
        np.testing.assert_array_equal(np.array([f for _, f in got]), golden_vectors[name + "_ratios"])
        for (o, _), want in zip(got, r["u8_sha256"]):
            assert sha(tiling.gather_u8(slide, o, r["patch"])) == want
        assert got[-1][1] >= 1.0 and all(f < 1.0 for _, f in got[:-1])
