"""`train(cfg)` -- drop-in for models/patch_cls_simple/train.py:59-301 (the training entry point).

Same config schema (config.yaml), same loop structure: per epoch 200 training steps
(train.py:142), `val_steps` validation steps drawn from the SAME sampler with the same
augmentations (train.py:198-204), `ReduceLROnPlateau(min, 0.1, patience 5)` on the validation
loss (train.py:120-122, 240), best-validation-accuracy checkpoint `out_dir/best_model.pth`
holding `model.state_dict()` (train.py:244-249).

What moved to the GPU: the step itself (forward, CrossEntropy, backward, Adam: HIP kernels,
`ResNet18HIP.train_step`), batch assembly (gather + /255 + NCHW + batch-level flips in
`dh_tile_gather_aug`), and the running loss / accuracy sums (accumulated on the device, read
once per epoch instead of four host syncs per step, train.py:174-180).

The test set is the reference's: `--extract_test` cuts `test.samples_per_class` JPEG patches per class
from the test slides into `test.dir` (`prepare_test_patches`, train.py:41-56), and every epoch ends
with the ImageFolder test loop (train.py:109-111, 253-283: class folders and files in torchvision's
sorted order, `ToTensor` = uint8 / 255 in float32, batches of `batch_size`, unshuffled) run through the
HIP model, and the loss / accuracy plots `loss.jpg` / `acc.jpg` (train.py:285-301) when matplotlib is
importable.  A missing `test.dir` skips the test loop (the reference would stop in ImageFolder).
The data source is any object with `device_batches(batch_size,
n_batches)`: `AnnoRegionRndSampler` over `cfg["dataset"]["folder"]` when that folder exists (as
in the reference, train.py:93-103; `.psi` images need the third-party psimage package), else a
closed-form synthetic slide with rectangular regions (`RectRegionRndSampler`, BASELINE configs[1]).
"""
from __future__ import annotations

import argparse
from pathlib import Path

import torch

from . import utils
from .ddp import broadcast_state_
from .model import ce_loss, get_model
from ...patch_samplers.region_samplers import (AnnoRegionRndSampler, RectRegionRndSampler, extract_and_save_subset,
                                               synthetic_regions)


def _rank_world():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _synthetic_sampler(cfg, device):
    from ... import tiles

    side = int(cfg.get("runtime", {}).get("synthetic_slide", 8192))
    slide = tiles.synth_slide(side, side, 0, device)
    regions = synthetic_regions(side, side, cfg["model"]["n_classes"], seed=0)
    return RectRegionRndSampler(slide, regions, layer=cfg["dataset"]["layer"], patch_size=cfg["dataset"]["patch_size"],
                                patches_from_one_region=cfg["dataset"]["patches_from_one_region"], seed=_rank_world()[0],
                                device=device)   # data parallel: every rank draws its own stream of patches


def prepare_test_patches(cfg, img_anno_paths=None, device="cuda"):
    """train.py:41-56: (re)create `test.dir` with `test.samples_per_class` JPEG patches per class from the test slides."""
    import shutil

    if img_anno_paths is None:
        img_anno_paths = utils.get_img_ano_paths(ds_folder=Path(cfg["dataset"]["folder"]), sample="test")
    out_dir = Path(cfg["test"]["dir"])
    if out_dir.exists() and out_dir.is_dir():
        shutil.rmtree(out_dir)
    return extract_and_save_subset(img_anno_paths=img_anno_paths, out_folder=out_dir, patch_size=cfg["dataset"]["patch_size"],
                                   layer=cfg["dataset"]["layer"], patches_per_class=cfg["test"]["samples_per_class"],
                                   device=device)


class TestImageFolder:
    """The reference's `ImageFolder(root, transform=ToTensor())` + `DataLoader(batch_size, shuffle=False)` (train.py:109-111,
    253-257) without torchvision: classes = sub-folders in sorted (string) order, samples = the image files of each class
    in sorted order, a sample = uint8 HWC / 255 as float32 CHW.  The decoded patches stay on the device as uint8."""

    EXT = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")

    def __init__(self, root, device):
        import numpy as np
        from PIL import Image

        root = Path(root)
        self.classes = sorted(d.name for d in root.iterdir() if d.is_dir())
        if not self.classes:
            raise FileNotFoundError(f"Couldn't find any class folder in {root}.")
        imgs, labels = [], []
        for ci, c in enumerate(self.classes):
            for f in sorted(q for q in (root / c).rglob("*") if q.is_file() and q.suffix.lower() in self.EXT):
                with Image.open(f) as im:
                    imgs.append(np.asarray(im.convert("RGB")))
                labels.append(ci)
        if not imgs:
            raise FileNotFoundError(f"Found no valid file for the classes {self.classes}.")
        self.u8 = torch.from_numpy(np.stack(imgs)).to(device)          # [N, P, P, 3]
        self.labels = torch.tensor(labels, dtype=torch.int64, device=device)
        # ToTensor divides on the CPU (correctly rounded); a float division on the GPU may differ in the last bit, so the 256
        # possible values come from a table computed on the host
        self._lut = torch.from_numpy(np.arange(256, dtype=np.float32) / np.float32(255)).to(device)

    def __len__(self):
        return int(self.labels.numel())

    def batches(self, batch_size: int):
        for b0 in range(0, len(self), batch_size):
            x = self._lut[self.u8[b0:b0 + batch_size].permute(0, 3, 1, 2).long()].contiguous()             # ToTensor
            yield x, self.labels[b0:b0 + batch_size]


def save_plot(out_dir, train_values, val_values, test_values, title, filename):
    """train.py:28-38; skipped when matplotlib is not importable."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:
        return False
    plt.figure()
    plt.plot(train_values, label="train")
    plt.plot(val_values, label="val")
    plt.plot(test_values, label="test")
    plt.title(title)
    plt.xlabel("Epoch")
    plt.legend()
    plt.savefig(Path(out_dir) / filename)
    plt.close()
    return True


class _PlateauLR:
    """ReduceLROnPlateau(mode="min", factor=0.1, patience=5) (train.py:120-122) for the fused step."""

    def __init__(self, lr, factor=0.1, patience=5, threshold=1e-4):
        self.lr, self.factor, self.patience, self.threshold = lr, factor, patience, threshold
        self.best, self.bad = float("inf"), 0

    def step(self, metric):
        if metric < self.best * (1 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
            if self.bad > self.patience:
                self.lr, self.bad = self.lr * self.factor, 0
        return self.lr


def _mean_over_ranks(*values):
    """Epoch metrics averaged over the ranks, so that every replica takes the same learning-rate and checkpoint decisions."""
    import torch.distributed as dist

    if _rank_world()[1] == 1:
        return values
    t = torch.tensor(values, dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t)
    return tuple((t / _rank_world()[1]).tolist())


def train(cfg, sampler=None, epochs=None, steps_per_epoch=200, log=print, model=None):
    """`model` / `sampler`: injected objects with `train_step(x, labels, lr=)` / `device_batches(bs, n, flips=)` (tests of the
    multi-process wiring run stubs on CPU tensors); by default the HIP model and the GPU samplers, which need a GPU."""
    device = utils.get_device()
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())   # under torch.distributed.run: the rank's own GPU
    rank, world = _rank_world()
    if rank != 0:
        log = lambda *_a, **_k: None   # noqa: E731  (rank 0 reports, as it alone writes the files)
    log(f"Using device: {device}" + (f"  ({world} ranks, data parallel)" if world > 1 else ""))
    if device.type != "cuda" and model is None:
        raise RuntimeError("deephisto_amd trains on the GPU only (HIP kernels); no CPU fallback")
    Path(cfg["training"]["save_dir"]).mkdir(parents=True, exist_ok=True)
    out_dir = Path(cfg["training"]["out_dir"])
    out_dir.mkdir(parents=True, exist_ok=True)

    if sampler is None:
        folder = Path(cfg["dataset"]["folder"])
        if world > 1:   # the annotation samplers draw from the global NumPy RNG (as the reference's do): one stream per rank
            import numpy as np
            np.random.seed(20240 + rank)
        if folder.exists():   # the reference's data source (train.py:93-103); .psi files need the psimage package
            sampler = AnnoRegionRndSampler(utils.get_img_ano_paths(folder, sample="train"),
                                           patch_size=cfg["dataset"]["patch_size"], layer=cfg["dataset"]["layer"],
                                           patches_from_one_region=cfg["dataset"]["patches_from_one_region"],
                                           one_image_for_batch=cfg["training"].get("one_image_for_batch", False),
                                           device=device)
        else:
            sampler = _synthetic_sampler(cfg, device)

    bs = cfg["training"]["batch_size"]
    # `model.arch: resnet50` selects the backbone of BASELINE configs[4] (bf16 engine); under torchrun (one process per GPU)
    # train_step averages the gradients over the ranks (bucketed all-reduce overlapped with backward) and rank 0 writes the files
    if model is None:
        model = get_model(cfg["model"]["n_classes"], cfg.get("runtime", {}).get("compute_dtype", "f32"),
                          arch=cfg["model"].get("arch", "resnet18")).to(device)
    if world > 1:   # every rank initialised its own weights: start all replicas from rank 0's (torch DDP's constructor broadcast)
        broadcast_state_(model)
    sched = _PlateauLR(cfg["training"]["lr"])
    history = {"train_loss": [], "train_acc": [], "val_loss": [], "val_acc": [], "lr": [], "test_loss": [], "test_acc": []}
    best_val_acc = 0.0
    test_dir = Path(cfg.get("test", {}).get("dir", "")) if cfg.get("test", {}).get("dir") else None
    test_set = TestImageFolder(test_dir, device) if test_dir is not None and test_dir.is_dir() else None
    if test_set is None:
        log("no test folder (run with --extract_test to create it): the test loop is skipped")
    n_epochs = epochs if epochs is not None else cfg["training"]["n_epochs"]

    for epoch in range(n_epochs):
        model.train()
        loss_sum = torch.zeros((), device=device)
        correct = torch.zeros((), device=device, dtype=torch.int64)
        total = 0
        for x, labels, _ in sampler.device_batches(bs, steps_per_epoch, flips=True):
            loss, logits = model.train_step(x, labels, lr=sched.lr)   # HIP: fwd + CE + bwd + Adam
            loss_sum += loss
            correct += (logits.argmax(1) == labels).sum()
            total += labels.numel()
        train_loss, train_acc = _mean_over_ranks(float(loss_sum) / steps_per_epoch, int(correct) / total)
        log(f"Epoch {epoch + 1}/{n_epochs}  Train Loss: {train_loss:.4f}, Train Acc: {train_acc:.4f}")

        # validation: same sampler, same augmentations, eval-mode BN, no update (train.py:190-236)
        if world > 1:   # per-rank batch statistics: validation, the checkpoint and the next epoch use rank 0's running statistics
            broadcast_state_(model, buffers_only=True)
        model.eval()
        val_steps = cfg["training"]["val_steps"]
        vloss = torch.zeros((), device=device)
        vcorrect, vtotal = torch.zeros((), device=device, dtype=torch.int64), 0
        for x, labels, _ in sampler.device_batches(bs, val_steps, flips=True):
            logits = model(x)
            vloss += ce_loss(logits, labels)                          # dh_ce_loss: CrossEntropyLoss(mean), train.py:117
            vcorrect += (logits.argmax(1) == labels).sum()
            vtotal += labels.numel()
        val_loss, val_acc = _mean_over_ranks(float(vloss) / val_steps, int(vcorrect) / vtotal)
        log(f"Val Loss: {val_loss:.4f}, Val Acc: {val_acc:.4f}")
        lr = sched.step(val_loss)
        log(f"Current Learning Rate: {lr:.6f}")
        if val_acc > best_val_acc:
            best_val_acc = val_acc
            if rank == 0:
                torch.save(model.state_dict(), out_dir / "best_model.pth")
        for k, v in zip(("train_loss", "train_acc", "val_loss", "val_acc", "lr"), (train_loss, train_acc, val_loss, val_acc, lr)):
            history[k].append(v)

        # test loop over the JPEG ImageFolder (train.py:251-283), plots (train.py:285-301)
        if test_set is not None:
            tloss = torch.zeros((), device=device)
            tcorrect, nb = torch.zeros((), device=device, dtype=torch.int64), 0
            for x, labels in test_set.batches(bs):
                logits = model(x)
                tloss += ce_loss(logits, labels)
                tcorrect += (logits.argmax(1) == labels).sum()
                nb += 1
            test_loss, test_acc = float(tloss) / nb, int(tcorrect) / len(test_set)
            history["test_loss"].append(test_loss)
            history["test_acc"].append(test_acc)
            log(f"Test Loss: {test_loss:.4f}, Test Acc: {test_acc:.4f}")
            if rank == 0:
                save_plot(out_dir, history["train_loss"], history["val_loss"], history["test_loss"], "Loss", "loss.jpg")
                save_plot(out_dir, history["train_acc"], history["val_acc"], history["test_acc"], "Acc", "acc.jpg")
    return model, history


def main(argv=None):
    """`python -m models.patch_cls_simple.train [--extract_test]` (train.py:304-315); `--config` / `--epochs` are
    additions.  The reference reads ./models/patch_cls_simple/config.yaml relative to the working directory;
    that file is used when it exists, else the config.yaml next to this module."""
    parser = argparse.ArgumentParser()
    parser.add_argument("--extract_test", action="store_true", default=False)
    parser.add_argument("--config", default=None)
    parser.add_argument("--epochs", type=int, default=None)
    parser.add_argument("--steps_per_epoch", type=int, default=200)   # train.py:142 hard-codes 200
    args = parser.parse_args(argv)
    if args.config is not None:
        cfg_path = Path(args.config)
    else:
        cwd_cfg = Path("./models/patch_cls_simple/config.yaml")
        cfg_path = cwd_cfg if cwd_cfg.exists() else Path(__file__).with_name("config.yaml")
    cfg = utils.load_config(cfg_path)
    # one process per GPU under torch.distributed.run: bind the rank's GPU and join the RCCL group before any other GPU call
    from ...distributed import finalize, init_from_env
    rank, world, _dev, owned = init_from_env()
    ok = False
    try:
        if args.extract_test and rank == 0:
            if Path(cfg["dataset"]["folder"]).exists():
                prepare_test_patches(cfg, device=utils.get_device())        # train.py:312-313
            else:
                print(f"--extract_test: dataset folder {cfg['dataset']['folder']} does not exist (synthetic training data): "
                      "no test patches are cut")
        if world > 1:
            import torch.distributed as dist
            dist.barrier()   # the test folder exists before any rank opens it
        out = train(cfg, epochs=args.epochs, steps_per_epoch=args.steps_per_epoch)
        ok = True
        return out
    finally:
        finalize(owned, ok)   # a failing rank leaves without a barrier: the launcher tears the job down


if __name__ == "__main__":
    main()
