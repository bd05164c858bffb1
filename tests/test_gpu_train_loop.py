"""GPU: the a9 batch contract (region sampler -> training batches) and the `train(cfg)` mirror."""
import numpy as np
import pytest
import torch

from oracle import resnet18 as oracle_net
from oracle import synth, tiling

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(built_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_region_sampler_output_contract(dev):
    """features f32[B,P,P,3] = u8/255 (bit-exact), labels int64[B], coords f32[B,2] = (pos_y,pos_x)
    -- patch_samplers/region_samplers.py:616-621, 729-738 -- and the overlap constraint."""
    from deephisto_amd.patch_samplers.region_samplers import RectRegion, RectRegionRndSampler
    host = synth.synth_slide(1500, 1700, 6)
    regions = [RectRegion("TUM", 100, 200, 900, 1000), RectRegion("AT", 600, 900, 1400, 1650), RectRegion("BG", 0, 0, 300, 260)]
    P = 224
    smp = RectRegionRndSampler(host, regions, layer=1, patch_size=P, seed=3, device=dev)
    assert smp.classes == ["AT", "BG", "TUM"]
    seen = set()
    for feats, labels, coords in smp.torch_generator(batch_size=16, n_batches=3):
        assert feats.dtype == torch.float32 and tuple(feats.shape) == (16, P, P, 3)
        assert labels.dtype == torch.int64 and tuple(labels.shape) == (16,)
        assert coords.dtype == torch.float32 and tuple(coords.shape) == (16, 2)
        yx = coords.cpu().numpy().astype(np.int64)
        want = np.stack([host[y:y + P, x:x + P] for y, x in yx]).astype(np.float32) / 255
        assert np.array_equal(feats.cpu().numpy().view(np.uint32), want.view(np.uint32))
        for (y, x), l in zip(yx, labels.cpu().numpy()):
            seen.add(int(l))
            ok = False
            for r in regions:
                if smp.classes.index(r.cls) != l:
                    continue
                oy = max(0, min(y + P, r.y1) - max(y, r.y0)); ox = max(0, min(x + P, r.x1) - max(x, r.x0))
                ok |= oy * ox >= 0.75 * P * P - 1 or r.area < P * P
            assert ok, (y, x, l)
    assert seen == {0, 1, 2}
    # user transforms are applied to the stacked features, like the reference
    tr = lambda f: f.permute(0, 3, 1, 2).contiguous()
    f, _, _ = next(smp.torch_generator(4, 1, transforms=tr))
    assert tuple(f.shape) == (4, 3, P, P)


def test_rect_sampler_device_batches_one_staged_copy(dev):
    """device_batches of the rectangle sampler: origins, labels and float coordinates travel as ONE staged copy per batch (round 5); the batch must
    equal what the same seed gives through sample_origins + the two coins + the gather kernel, labels and coordinates included."""
    from deephisto_amd import tiles
    from deephisto_amd._lib import DH_LAYOUT_NCHW
    from deephisto_amd.patch_samplers.region_samplers import RectRegion, RectRegionRndSampler
    host = synth.synth_slide(1200, 1400, 9)
    regions = [RectRegion("TUM", 100, 200, 900, 1000), RectRegion("AT", 500, 700, 1150, 1350), RectRegion("BG", 0, 0, 300, 260)]
    P, B = 96, 24
    a = RectRegionRndSampler(host, regions, layer=1, patch_size=P, seed=11, device=dev)
    b = RectRegionRndSampler(host, regions, layer=1, patch_size=P, seed=11, device=dev)
    for x, lab, c in a.device_batches(B, 5, flips=True):
        yx, want_lab = b.sample_origins(B)
        fh = bool(b._rng.random() < 0.5)
        fv = bool(b._rng.random() < 0.5)
        want = tiles.gather_tiles_aug(a.slide, torch.from_numpy(yx).to(dev), P, DH_LAYOUT_NCHW, torch.float32, fh, fv)
        assert torch.equal(x, want)
        assert lab.dtype == torch.int64 and np.array_equal(lab.cpu().numpy(), want_lab)
        assert c.dtype == torch.float32 and np.array_equal(c.cpu().numpy(), yx.astype(np.float32))


def test_gather_aug_flips_match_torch(dev):
    from deephisto_amd import tiles
    from deephisto_amd._lib import DH_LAYOUT_NCHW, DH_LAYOUT_NHWC
    host = synth.synth_slide(400, 500, 2)
    slide = torch.from_numpy(host).to(dev)
    o = torch.tensor([[3, 7], [100, 200], [144, 244]], dtype=torch.int32, device=dev)
    base = tiles.gather_tiles(slide, o, 96, DH_LAYOUT_NCHW, torch.float32, check_bounds=False)
    for fh in (False, True):
        for fv in (False, True):
            got = tiles.gather_tiles_aug(slide, o, 96, DH_LAYOUT_NCHW, torch.float32, fh, fv)
            want = base
            if fh: want = torch.flip(want, dims=[3])   # RandomHorizontalFlip on [B,C,H,W]
            if fv: want = torch.flip(want, dims=[2])
            assert torch.equal(got, want)
    got = tiles.gather_tiles_aug(slide, o, 96, DH_LAYOUT_NHWC, torch.bfloat16, True, False)
    want = torch.flip(base, dims=[3]).permute(0, 2, 3, 1).to(torch.bfloat16)
    assert torch.equal(got, want)


def test_train_mirror_runs_and_learns(dev, tmp_path):
    """train(cfg) on a synthetic annotated slide whose classes are separable (constant-colour
    regions): loss falls, checkpoint is written and reloads into the oracle's module layout."""
    from deephisto_amd.models.patch_cls_simple.train import train
    from deephisto_amd.patch_samplers.region_samplers import RectRegion, RectRegionRndSampler
    from oracle import resnet18 as oracle_net
    side = 1024
    host = synth.synth_slide(side, side, 1) // 4            # dim noise
    regions = []
    for i, name in enumerate(["AT", "BG", "LP", "MM", "TUM"]):
        y0 = i * 200
        host[y0:y0 + 200, :, :] += np.array([40 * i, 200 - 40 * i, 20 * i], dtype=np.uint8)   # class colour
        regions.append(RectRegion(name, y0, 0, y0 + 200, side))
    smp = RectRegionRndSampler(host, regions, layer=1, patch_size=64, seed=0, device=dev)
    cfg = {"model": {"n_classes": 5},
           "training": {"batch_size": 16, "n_epochs": 2, "lr": 1e-3, "save_dir": str(tmp_path / "ck"),
                        "out_dir": str(tmp_path / "out"), "val_steps": 2},
           "dataset": {"folder": "/nonexistent", "layer": 1, "patch_size": 64, "patches_from_one_region": 4}}
    torch.manual_seed(0)
    model, hist = train(cfg, sampler=smp, epochs=2, steps_per_epoch=40, log=lambda *a: None)
    assert hist["train_loss"][1] < hist["train_loss"][0]
    assert hist["train_acc"][1] > 0.3   # 5 classes: chance is 0.2
    ck = torch.load(tmp_path / "out" / "best_model.pth", weights_only=True)
    ref = oracle_net.ResNet18Oracle(5)
    ref.load_state_dict(ck)            # same keys / shapes as torchvision's layout
    x = torch.rand(2, 3, 64, 64)
    with torch.no_grad():
        want = ref.eval()(x)
    model.load_state_dict(ck)          # the checkpoint is the BEST epoch, the returned model the last one
    got = model.eval()(x.to(dev)).cpu()
    assert float((got - want).abs().max()) <= 2e-4


def test_anno_region_sampler_device_batches(dev):
    """AnnoRegionRndSampler (polygon annotations, two images): torch_generator keeps the reference's output
    contract bit-exactly (uint8/255 in float32), device_batches = permute + batch-level flips, and the fused
    train step runs on its batches."""
    from deephisto_amd.patch_samplers.region_samplers import AnnoRegionRndSampler
    rng = np.random.default_rng(0)
    img0 = synth.synth_slide(1500, 1800, 3)
    img1 = synth.synth_slide(1400, 1300, 4)
    ang = np.sort(rng.uniform(0, 2 * np.pi, 15))
    star = np.stack([800 + rng.uniform(300, 600, 15) * np.cos(ang), 750 + rng.uniform(300, 600, 15) * np.sin(ang)], 1)
    a0 = [{"class": "TUM", "vertices": star.tolist()}, {"class": "BG", "vertices": [[50, 50], [700, 60], [680, 700], [40, 650]]}]
    a1 = [{"class": "LP", "vertices": [[100, 100], [1200, 150], [1100, 1300], [150, 1200]]}]
    smp = AnnoRegionRndSampler([(img0, a0), (img1, a1)], layer=1, patch_size=96, patches_from_one_region=4, device=dev)
    imgs = [img0, img1]
    np.random.seed(21)
    recs = smp._records(2 * 8)
    np.random.seed(21)
    out = list(smp.torch_generator(batch_size=8, n_batches=2))
    assert len(out) == 2
    for b, (f, lab, c) in enumerate(out):
        assert f.dtype == torch.float32 and tuple(f.shape) == (8, 96, 96, 3) and f.device.type == "cuda"
        assert lab.dtype == torch.int64 and c.dtype == torch.float32 and tuple(c.shape) == (8, 2)
        for i, (j, y, x, cls) in enumerate(recs[8 * b:8 * b + 8]):
            want = imgs[j][y:y + 96, x:x + 96].astype(np.float32) / 255
            np.testing.assert_array_equal(f[i].cpu().numpy(), want)
            assert int(lab[i]) == cls and c[i].tolist() == [float(y), float(x)]
    # device_batches: NCHW + the two torch coins per batch (horizontal first)
    np.random.seed(22); torch.manual_seed(5)
    recs = smp._records(8)
    np.random.seed(22); torch.manual_seed(5)
    fh = torch.rand(1).item() < 0.5
    fv = torch.rand(1).item() < 0.5
    torch.manual_seed(5)
    x, lab, c = next(smp.device_batches(8, 1))
    assert tuple(x.shape) == (8, 3, 96, 96)
    for i, (j, y, xx, cls) in enumerate(recs):
        want = torch.from_numpy(imgs[j][y:y + 96, xx:xx + 96].astype(np.float32) / 255).permute(2, 0, 1)
        if fh: want = torch.flip(want, dims=[2])
        if fv: want = torch.flip(want, dims=[1])
        assert torch.equal(x[i].cpu(), want)
    # endless single-sample dataset
    np.random.seed(23)
    it = iter(smp.torch_iterable_dataset())
    f0, l0, c0 = next(it)
    assert tuple(f0.shape) == (96, 96, 3) and l0.dtype == torch.int64 and tuple(c0.shape) == (2,)
    y0, x0 = int(c0[0]), int(c0[1])
    assert any(torch.equal(f0.cpu(), torch.from_numpy(im[y0:y0 + 96, x0:x0 + 96].astype(np.float32) / 255))
               for im in imgs if y0 + 96 <= im.shape[0] and x0 + 96 <= im.shape[1])
    # and the training step consumes it
    from deephisto_amd.models.patch_cls_simple.model import get_model
    torch.manual_seed(0)
    model = get_model(3, "f32").to(dev).train()
    losses = []
    for xb, yb, _ in smp.device_batches(8, 6):
        loss, _ = model.train_step(xb, yb, lr=1e-3)
        losses.append(float(loss))
    assert all(np.isfinite(losses))


def test_border_regions_zero_filled_on_device(dev):
    """Regions touching / leaving the image border: the device gather writes 0 for pixels outside the
    slide and reads nothing outside it; batches equal the host records of tests/test_region_borders.py."""
    from deephisto_amd.patch_samplers.region_samplers import AnnoRegionDenseSampler, AnnoRegionRndSampler
    img = synth.synth_slide(700, 1000, 5)

    def expected(y, x, ps):
        out = np.zeros((ps, ps, 3), np.uint8)
        ya, yb, xa, xb = max(y, 0), min(y + ps, 700), max(x, 0), min(x + ps, 1000)
        if yb > ya and xb > xa:
            out[ya - y:yb - y, xa - x:xb - x] = img[ya:yb, xa:xb]
        return out.astype(np.float32) / 255

    anno = [{"class": "TUM", "vertices": [[790, 50], [1000, 50], [1000, 650], [790, 650]]},
            {"class": "BG", "vertices": [[-50, -30], [400, -30], [400, 380], [-50, 380]]}]
    smp = AnnoRegionRndSampler([(img, anno)], layer=1, patch_size=256, region_intersection=0.5, device=dev)
    np.random.seed(3)
    recs = smp._records(32)
    assert any(x + 256 > 1000 for _, _, x, _ in recs) and any(y < 0 or x < 0 for _, y, x, _ in recs)
    np.random.seed(3)
    f, lab, c = next(smp.torch_generator(batch_size=32, n_batches=1))
    for i, (_, y, x, cls) in enumerate(recs):
        np.testing.assert_array_equal(f[i].cpu().numpy(), expected(y, x, 256))
    dense = AnnoRegionDenseSampler([(img, anno)], layer=1, patch_size=128, stride=64, device=dev)
    n = 0
    for x, lab, c in dense.device_batches(16, layout=0):
        for i in range(x.shape[0]):
            np.testing.assert_array_equal(x[i].cpu().numpy(), expected(int(c[i, 0]), int(c[i, 1]), 128))
            n += 1
    assert n > 0


def test_reference_caller_shape_through_aliases(dev, tmp_path):
    """The dense branch of the reference's `__main__` (examples/predict_full_patched.py:128-177) written against
    the reference's module paths: load_model from a state_dict file, FullImageDenseSampler(img, layer, patch_size,
    batch_size, stride), ImagePredictorPatched(..., patch_sampler=sampler.generator(), batch_predictor=lambda ...),
    process(), perform_and_save_visualizations."""
    import deephisto_amd
    deephisto_amd.install_aliases(force=True)
    try:
        from anno.utils import AnnoDescription
        from examples.predict_full_patched import (ImagePredictorPatched, batch_predictor, load_model,
                                                   perform_and_save_visualizations)
        from models.patch_cls_simple import utils
        from models.patch_cls_simple.model import get_model
        from patch_samplers.full_samplers import FullImageDenseSampler

        ref = oracle_net.seeded_model(2, 5, perturb_bn=True).eval()
        torch.save(ref.state_dict(), tmp_path / "best_model.pth")
        img_path = tmp_path / "test_01.npy"
        host = synth.synth_slide(900, 1100, 8)
        np.save(img_path, host)

        device = utils.get_device()
        model = load_model(tmp_path / "best_model.pth", device)
        anno_dsc = AnnoDescription.with_known_colors({"AT": (245, 119, 34), "BG": (153, 255, 255), "LP": (64, 170, 72),
                                                      "MM": (255, 0, 0), "TUM": (33, 67, 156)})
        layer, downscale_vis = 1, 16
        patch_sampler = FullImageDenseSampler(img_path, layer=layer, patch_size=224, batch_size=8, stride=112)
        predictor = ImagePredictorPatched(img_path, patch_sampler=patch_sampler.generator(),
                                          batch_predictor=lambda patches: batch_predictor(patches, model, device),
                                          anno=anno_dsc, layer=layer, downscale=downscale_vis)
        pred = predictor.process()
        assert pred.dtype == np.int64 and pred.shape == (900 // 16, 1100 // 16)
        # against the oracle pipeline on the same tiles
        o = tiling.batched_origins(900, 1100, 224, 112, 8).reshape(-1, 2)
        with torch.no_grad():
            logits = ref(torch.from_numpy(tiling.features_nchw_predictor(host, o, 224))).numpy()
        canvas = tiling.accumulate_logits(900, 1100, 5, 16, 224, o, logits)
        want = tiling.class_map(canvas)
        top2 = np.sort(canvas, axis=2)
        decided = (top2[..., -1] - top2[..., -2]) > 1e-3
        assert np.array_equal(pred[decided], want[decided])
        mask, small, ov = perform_and_save_visualizations(img_path, anno_dsc, pred, out_dir=tmp_path / "output")
        assert (tmp_path / "output" / "test_01_mask.jpg").exists() and mask.shape == pred.shape + (3,)
        assert get_model(5).__class__.__name__ == model.__class__.__name__
    finally:
        deephisto_amd.uninstall_aliases()


def test_train_mirror_resnet50_bf16(dev, tmp_path):
    """The same entry point with `model.arch: resnet50` (BASELINE configs[4]: bf16 engine): learns the separable synthetic
    classes, the checkpoint has torchvision's ResNet-50 layout and evaluates like the oracle holding it (bf16 tolerance)."""
    from deephisto_amd.models.patch_cls_simple.train import train
    from deephisto_amd.patch_samplers.region_samplers import RectRegion, RectRegionRndSampler
    from oracle import resnet50 as o50
    side = 1024
    host = synth.synth_slide(side, side, 1) // 4
    regions = []
    for i, name in enumerate(["AT", "BG", "LP", "MM", "TUM"]):
        y0 = i * 200
        host[y0:y0 + 200, :, :] += np.array([40 * i, 200 - 40 * i, 20 * i], dtype=np.uint8)
        regions.append(RectRegion(name, y0, 0, y0 + 200, side))
    smp = RectRegionRndSampler(host, regions, layer=1, patch_size=64, seed=0, device=dev)
    cfg = {"model": {"n_classes": 5, "arch": "resnet50"},
           "training": {"batch_size": 16, "n_epochs": 2, "lr": 1e-3, "save_dir": str(tmp_path / "ck"),
                        "out_dir": str(tmp_path / "out"), "val_steps": 2},
           "dataset": {"folder": "/nonexistent", "layer": 1, "patch_size": 64, "patches_from_one_region": 4}}
    torch.manual_seed(0)
    model, hist = train(cfg, sampler=smp, epochs=2, steps_per_epoch=40, log=lambda *a: None)
    assert hist["train_loss"][1] < hist["train_loss"][0] and hist["train_acc"][1] > 0.3
    ck = torch.load(tmp_path / "out" / "best_model.pth", weights_only=True)
    ref = o50.ResNet50Oracle(5)
    ref.load_state_dict(ck)
    x = torch.rand(4, 3, 64, 64)
    with torch.no_grad():
        want = ref.eval()(x)
    model.load_state_dict(ck)
    got = model.eval()(x.to(dev)).cpu()
    assert float((got - want).abs().max()) <= 5e-2 * max(1.0, float(want.abs().max()))


def test_extract_test_patches_and_test_loop(dev, tmp_path):
    """`--extract_test` path (train.py:41-56 + region_samplers.py:874-909) and the per-epoch ImageFolder test loop
    (train.py:109-111, 251-301): JPEG patches per class in `test.dir/<class index>/<n>.jpg`, every one of them from inside a
    region of its class (the classes are colour-coded, JPEG noise is far below the colour distance), then train() with that
    folder present reports test loss / accuracy per epoch and writes the two plots."""
    from PIL import Image
    from deephisto_amd.models.patch_cls_simple.train import TestImageFolder, prepare_test_patches, train
    from deephisto_amd.patch_samplers.region_samplers import RectRegion, RectRegionRndSampler
    side = 1200
    host = synth.synth_slide(side, side, 2) // 8
    colours = {"AT": (200, 30, 30), "BG": (30, 200, 30), "TUM": (30, 30, 200)}
    anno, regions = [], []
    for i, (name, col) in enumerate(colours.items()):
        y0, y1 = 40 + i * 380, 40 + i * 380 + 330
        host[y0:y1, 60:1100, :] += np.array(col, dtype=np.uint8)
        anno.append({"class": name, "vertices": [[60, y0], [1100, y0], [1100, y1], [60, y1]]})
        regions.append(RectRegion(name, y0, 60, y1, 1100))
    cfg = {"model": {"n_classes": 3},
           "training": {"batch_size": 8, "n_epochs": 1, "lr": 1e-3, "save_dir": str(tmp_path / "ck"),
                        "out_dir": str(tmp_path / "out"), "val_steps": 2},
           "test": {"dir": str(tmp_path / "test"), "samples_per_class": 12},
           "dataset": {"folder": "/nonexistent", "layer": 1, "patch_size": 64, "patches_from_one_region": 4}}
    (tmp_path / "test" / "stale").mkdir(parents=True)                     # an old folder is replaced, not merged
    np.random.seed(3)
    counts = prepare_test_patches(cfg, img_anno_paths=[(host, anno)], device=dev)
    assert counts == {"AT": 12, "BG": 12, "TUM": 12} and not (tmp_path / "test" / "stale").exists()
    for ci, (name, col) in enumerate(colours.items()):
        files = sorted((tmp_path / "test" / str(ci)).iterdir())
        assert [f.name for f in files] == sorted(f"{k}.jpg" for k in range(12))
        for f in files:
            im = np.asarray(Image.open(f))
            assert im.shape == (64, 64, 3)
            mean = im.reshape(-1, 3).mean(0)
            dist = {n: np.abs(mean - (np.array(c) + 15)).max() for n, c in colours.items()}   # 95 % of the patch inside a region
            if ci == 0:
                # the reference's `c_idx = cls_idx or random` (region_samplers.py:555, 576) cannot force class index 0:
                # folder "0" receives patches of randomly drawn classes there, and therefore here
                assert min(dist.values()) < 25
            else:
                assert dist[name] < 25
    ts = TestImageFolder(tmp_path / "test", dev)
    assert ts.classes == ["0", "1", "2"] and len(ts) == 36
    xb, yb = next(ts.batches(8))
    first = np.asarray(Image.open(tmp_path / "test" / "0" / "0.jpg")).astype(np.float32) / 255       # ToTensor
    assert xb.dtype == torch.float32 and tuple(xb.shape) == (8, 3, 64, 64) and yb.tolist() == [0] * 8
    np.testing.assert_array_equal(xb[0].cpu().numpy(), first.transpose(2, 0, 1))
    order = [int(v) for v in sorted(str(k) for k in range(12))]                                       # "0", "1", "10", "11", "2", ...
    second = np.asarray(Image.open(tmp_path / "test" / "0" / f"{order[2]}.jpg")).astype(np.float32) / 255
    np.testing.assert_array_equal(xb[2].cpu().numpy(), second.transpose(2, 0, 1))
    smp = RectRegionRndSampler(host, regions, layer=1, patch_size=64, seed=0, device=dev)
    lines = []
    torch.manual_seed(0)
    model, hist = train(cfg, sampler=smp, epochs=2, steps_per_epoch=30, log=lambda *a: lines.append(" ".join(map(str, a))))
    assert len(hist["test_loss"]) == 2 and all(np.isfinite(hist["test_loss"]))
    assert hist["test_acc"][1] > 0.45                                     # folders 1, 2 are learnable (folder 0 is mixed, see above)
    assert sum(l.startswith("Test Loss:") for l in lines) == 2
    assert (tmp_path / "out" / "loss.jpg").exists() and (tmp_path / "out" / "acc.jpg").exists()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_load_state_dict_after_fused_step_wins(dtype):
    """after `train_step` the library's masters are newer than the nn.Parameters; a
    `load_state_dict(sd0)` must not be undone by the lazy pull-back.  train_step -> load_state_dict(sd0) -> eval()(x)
    equals a fresh model holding sd0 (parameters AND running statistics), and training resumes from sd0."""
    from deephisto_amd.models.patch_cls_simple.model import get_model
    from oracle import resnet18 as o18
    dev = torch.device("cuda:0")
    sd0 = {k: v.clone() for k, v in o18.seeded_model(5, 5, perturb_bn=True).state_dict().items()}
    g = torch.Generator().manual_seed(9)
    x = torch.rand(8, 3, 64, 64, generator=g).to(dev)
    y = torch.randint(0, 5, (8,), generator=g).to(dev)

    fresh = get_model(5, dtype)
    fresh.load_state_dict(sd0)
    fresh.to(dev).eval()
    want = fresh(x).cpu()

    m = get_model(5, dtype)
    m.load_state_dict(sd0)
    m.to(dev).train()
    for _ in range(2):
        m.train_step(x, y, lr=1e-2)
    m.load_state_dict(sd0)
    got = m.eval()(x).cpu()
    assert torch.equal(got, want), f"max diff {float((got - want).abs().max())}"
    after = m.state_dict()
    for k, v in sd0.items():
        assert torch.equal(after[k].cpu(), v), k
    # training resumes from sd0: the first step's logits equal those of a fresh model's first step
    m.train()
    fresh.train()
    _, l1 = m.train_step(x, y, lr=1e-2)
    _, l2 = fresh.train_step(x, y, lr=1e-2)
    assert torch.equal(l1.cpu(), l2.cpu())
    # ... and so do the running statistics both then hold
    a, b = m.state_dict(), fresh.state_dict()
    for k in a:
        if "running" in k:
            assert torch.equal(a[k].cpu(), b[k].cpu()), k
