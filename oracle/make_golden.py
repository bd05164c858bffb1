#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE REFERENCE's own sampler / predictor.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Runs only in the build container
(needs the read-only reference tree, default /root/reference); the fixtures it
writes are committed, the reference never travels.

The reference imports three third-party packages that are absent here
(`psimage`, `torchvision`, `distinctipy`).  None of them carries arithmetic that
these fixtures pin:
  * `psimage.PSImage` is the slide *file reader*; the stand-in below serves the
    same `uint8[h, w, 3]` region reads from an in-memory array, so every line of
    the reference that is pinned here (full_samplers.py:302-452 tile order,
    padding, progress, /255 features, coords; predict_full_patched.py:22-78
    accumulation, argmax, batch_predictor) is the reference's own code.
  * `torchvision` / `distinctipy` are only touched by import statements on this
    path (model factory / colour palette) -- empty modules satisfy them.  The
    ResNet-18 itself is therefore NOT pinned by this script (parity unpinned,
    see oracle/resnet18.py).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import sys
import types
from dataclasses import dataclass
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from oracle.synth import synth_slide  # noqa: E402

_SLIDES: dict[str, np.ndarray] = {}


class _ArrayPSImage:
    """In-memory stand-in for psimage.core.image.PSImage (reader only)."""

    def __init__(self, path):
        self._a = _SLIDES[str(path)]
        self.height, self.width = self._a.shape[:2]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def close(self):
        pass

    def _assert_layer(self, layer):
        assert layer >= 1

    def layer_size(self, layer):
        return self._a.shape[0], self._a.shape[1]

    def get_region_from_layer(self, layer, p0, p1):
        return self._a[p0[0]:p1[0], p0[1]:p1[1], :]


@dataclass
class _Patch:
    layer: int
    pos_x: int
    pos_y: int
    patch_size: int
    data: np.ndarray


def _install_standins():
    ps = types.ModuleType("psimage")
    core = types.ModuleType("psimage.core")
    img = types.ModuleType("psimage.core.image")
    pat = types.ModuleType("psimage.core.patches")
    img.PSImage = _ArrayPSImage
    pat.Patch = _Patch
    ps.core, core.image, core.patches = core, img, pat
    sys.modules.update({"psimage": ps, "psimage.core": core,
                        "psimage.core.image": img, "psimage.core.patches": pat})
    for name in ("distinctipy", "torchvision", "torchvision.models"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class _ShapeOnlyDense:
    """Run the reference's _create_batched_coords for a (h, w) without any pixels."""

    def __new__(cls, ref_cls, h, w, patch, stride, batch):
        o = object.__new__(ref_cls)
        o.h, o.w, o.patch_size, o.stride, o.batch_size = h, w, patch, stride, batch
        return o


def toy_logits_from_patches(patches) -> np.ndarray:
    """Deterministic closed-form 'model' used as the batch_predictor callback:
    5 logits per tile from integer pixel sums (float32, order-independent inputs)."""
    out = np.empty((len(patches), 5), dtype=np.float32)
    for i, p in enumerate(patches):
        s = p.data.reshape(-1, 3).astype(np.int64).sum(axis=0)  # exact
        r, g, b = (int(v) for v in s)
        n = p.data.shape[0] * p.data.shape[1]
        out[i] = np.array([r - g, g - b, b - r, (r + g + b) - 382 * n, (r ^ g ^ b) % 1001 - 500],
                          dtype=np.float64).astype(np.float32) / np.float32(n)
    return out


def toy_torch_model(seed: int = 7) -> torch.nn.Module:
    g = torch.Generator().manual_seed(seed)
    m = torch.nn.Sequential(
        torch.nn.Conv2d(3, 8, 5, 4, 2), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(1),
        torch.nn.Flatten(), torch.nn.Linear(8, 5))
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    return m.eval()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=str(REPO / "tests" / "golden"))
    args = ap.parse_args()
    out = Path(args.out)
    out.mkdir(parents=True, exist_ok=True)

    _install_standins()
    sys.path.insert(0, args.reference)
    from patch_samplers.full_samplers import FullImageDenseSampler, FullImageRndSampler, SamplerExecutionMode
    from examples.predict_full_patched import ImagePredictorPatched, batch_predictor

    meta: dict = {"generator": "oracle/make_golden.py", "reference": "xubiker/deephisto@2025-02-22"}

    # ---- (1) a1: tile grids, straight from the reference's _create_batched_coords
    grids = {}
    arrays = {}
    for name, (h, w, p, s, b) in {
        "g4096_256_256_64": (4096, 4096, 256, 256, 64),
        "g4096_224_112_16": (4096, 4096, 224, 112, 16),
        "g1000x1300_256_256_16": (1000, 1300, 256, 256, 16),
        "g256x600_256_256_4": (256, 600, 256, 256, 4),
        "g600x256_256_256_4": (600, 256, 256, 256, 4),
        "g1000x1300_224_112_64": (1000, 1300, 224, 112, 64),
        "g777x1033_100_37_7": (777, 1033, 100, 37, 7),
        "g50000_256_256_64": (50000, 50000, 256, 256, 64),
        "g50000_224_112_64": (50000, 50000, 224, 112, 64),
    }.items():
        smp = _ShapeOnlyDense(FullImageDenseSampler, h, w, p, s, b)
        cb = smp._create_batched_coords()
        flat = np.array([c for batch in cb for c in batch], dtype=np.int32)
        # count of padded duplicates = copies of the corner at the tail beyond the first
        corner = flat[-1]
        k = 0
        while k + 1 < len(flat) and (flat[-2 - k] == corner).all():
            k += 1
        n_unique = len(flat) - k
        grids[name] = {"h": h, "w": w, "patch": p, "stride": s, "batch": b,
                       "n_batches": len(cb), "n_padded": int(len(flat)), "n_unique": int(n_unique),
                       "sha256_int32_yx_padded": sha(flat)}
        if len(flat) <= 4096:
            arrays[name] = flat
        else:
            arrays[name + "_head"] = flat[:64]
            arrays[name + "_tail"] = flat[-64:]
    meta["grids"] = grids
    np.savez_compressed(out / "grids.npz", **arrays)

    # ---- (2) a2-a4: generator() / generator_torch() on closed-form slides
    feats = {}
    farr = {}
    for name, (h, w, p, s, b, seed) in {
        "f1024_256_256_4": (1024, 1024, 256, 256, 4, 0),
        "f600x900_224_112_16": (600, 900, 224, 112, 16, 1),
    }.items():
        _SLIDES[name] = synth_slide(h, w, seed)
        smp = FullImageDenseSampler(name, layer=1, patch_size=p, batch_size=b,
                                    mode=SamplerExecutionMode.INMEMORY_SINGLEPROC, stride=s)
        rec = {"h": h, "w": w, "patch": p, "stride": s, "batch": b, "seed": seed,
               "feature_sha256": [], "u8_sha256": [], "progress": [], "is_view": True}
        coords_all = []
        for (patches, prog), (f, c, prog2) in zip(smp.generator(), smp.generator_torch()):
            assert prog == prog2
            rec["progress"].append(prog)
            rec["is_view"] = rec["is_view"] and all(q.data.base is not None for q in patches)
            rec["u8_sha256"].append(sha(np.stack([q.data for q in patches])))
            assert f.dtype == torch.float32 and tuple(f.shape) == (b, p, p, 3)
            rec["feature_sha256"].append(sha(f.numpy()))
            coords_all.append(c.numpy())
            if len(rec["progress"]) == 1:
                farr[name + "_first_crop"] = f.numpy()[:, :8, :8, :].copy()
        farr[name + "_coords"] = np.stack(coords_all)
        feats[name] = rec
    meta["features"] = feats

    # ---- (3) a5/a8: ImagePredictorPatched.process with two batch predictors
    class _Anno:
        anno_classes = [0, 1, 2, 3, 4]

    preds = {}
    for name, (h, w, p, s, b, d, seed) in {
        "p1000x1300_256_256_16_d16": (1000, 1300, 256, 256, 16, 16, 2),
        "p600x900_224_112_16_d16": (600, 900, 224, 112, 16, 16, 1),
        "p700x500_100_60_8_d7": (700, 500, 100, 60, 8, 7, 3),
    }.items():
        _SLIDES[name] = synth_slide(h, w, seed)
        # (a) closed-form callback
        smp = FullImageDenseSampler(name, layer=1, patch_size=p, batch_size=b,
                                    mode=SamplerExecutionMode.INMEMORY_SINGLEPROC, stride=s)
        logs = []

        def cb(patches, _logs=logs):
            v = toy_logits_from_patches(patches)
            _logs.append(v)
            return v

        cmap = ImagePredictorPatched(name, smp.generator(), cb, _Anno(), layer=1, downscale=d).process()
        assert cmap.dtype == np.int64
        farr[name + "_toy_logits"] = np.concatenate(logs)
        farr[name + "_toy_map"] = cmap.astype(np.int16)
        # (b) the reference's own batch_predictor with a small seeded torch model on CPU
        model = toy_torch_model()
        smp = FullImageDenseSampler(name, layer=1, patch_size=p, batch_size=b,
                                    mode=SamplerExecutionMode.INMEMORY_SINGLEPROC, stride=s)
        logs2 = []

        def cb2(patches, _logs=logs2):
            v = batch_predictor(patches, model, torch.device("cpu"))
            assert v.dtype == np.float32
            _logs.append(v)
            return v

        cmap2 = ImagePredictorPatched(name, smp.generator(), cb2, _Anno(), layer=1, downscale=d).process()
        farr[name + "_torch_logits"] = np.concatenate(logs2)
        farr[name + "_torch_map"] = cmap2.astype(np.int16)
        preds[name] = {"h": h, "w": w, "patch": p, "stride": s, "batch": b, "downscale": d,
                       "seed": seed, "map_shape": list(cmap.shape), "toy_model_seed": 7}
    meta["predict"] = preds

    # ---- (4) next row f1: FullImageRndSampler under a fixed NumPy seed (full_samplers.py:21-299)
    rnd = {}
    for name, (h, w, p, b, dl, sp, seed, npseed) in {
        "r700x900_128_8": (700, 900, 128, 8, 2, 16, 5, 1234),
        "r512x640_96_4_dl1": (512, 640, 96, 4, 1, 16, 6, 99),
    }.items():
        _SLIDES[name] = synth_slide(h, w, seed)
        np.random.seed(npseed)
        smp = FullImageRndSampler(name, layer=1, patch_size=p, batch_size=b,
                                  mode=SamplerExecutionMode.INMEMORY_SINGLEPROC, dense_level=dl, speedup=sp)
        origins, ratios, u8 = [], [], []
        for patches, ratio in smp.generator():
            origins.append([(q.pos_y, q.pos_x) for q in patches])
            ratios.append(ratio)
            u8.append(sha(np.stack([q.data for q in patches])))
        farr[name + "_origins"] = np.array(origins, dtype=np.int32)
        farr[name + "_ratios"] = np.array(ratios, dtype=np.float64)
        np.random.seed(npseed)
        smp2 = FullImageRndSampler(name, layer=1, patch_size=p, batch_size=b,
                                   mode=SamplerExecutionMode.INMEMORY_SINGLEPROC, dense_level=dl, speedup=sp)
        f0, c0, _ = next(smp2.generator_torch())
        farr[name + "_torch_first_crop"] = f0.numpy()[:, :4, :4, :].copy()   # NOT divided by 255 (full_samplers.py:286)
        farr[name + "_torch_first_coords"] = c0.numpy()
        rnd[name] = {"h": h, "w": w, "patch": p, "batch": b, "dense_level": dl, "speedup": sp, "seed": seed,
                     "np_seed": npseed, "n_batches": len(ratios), "u8_sha256": u8,
                     "coords_dtype": str(c0.dtype), "features_dtype": str(f0.dtype)}
    meta["random_sampler"] = rnd

    np.savez_compressed(out / "vectors.npz", **farr)
    (out / "golden.json").write_text(json.dumps(meta, indent=1, sort_keys=True))
    print("wrote", sorted(q.name for q in out.iterdir()))


if __name__ == "__main__":
    main()
