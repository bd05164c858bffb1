"""GPU: the a9 batch contract (region sampler -> training batches) and the `train(cfg)` mirror."""
import numpy as np
import pytest
import torch

from oracle import synth

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
