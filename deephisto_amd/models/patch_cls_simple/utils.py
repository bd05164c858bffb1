"""Config loader, device picker and dataset listing of the training entry point.

Behaviour of models/patch_cls_simple/utils.py:1-17 (YAML -> dict; device preference mps, then
cuda, then cpu) and of the reference's top-level utils.py:4-14 (image / annotation pairs)."""
from __future__ import annotations

from pathlib import Path


def load_config(config_path) -> dict:
    """Parse the YAML training configuration (safe loader) into a plain dict."""
    import yaml

    return yaml.safe_load(Path(config_path).read_text())


def get_device():
    """First available of ("mps" when torch was built with it, "cuda", "cpu") -- the reference's
    preference order.  On PyTorch-ROCm an MI355X is reported as "cuda"."""
    import torch

    candidates = (("mps", torch.backends.mps.is_built), ("cuda", torch.cuda.is_available), ("cpu", lambda: True))
    return torch.device(next(name for name, available in candidates if available()))


def get_img_ano_paths(ds_folder, sample: str = "train") -> list[tuple[Path, Path]]:
    """(image, annotation) pairs of a dataset folder: every `images/<sample>/*.psi` with its
    `annotations/<sample>/<stem>.json`."""
    root = Path(ds_folder)
    images = sorted(p for p in (root / "images" / sample).iterdir() if p.is_file() and p.suffix == ".psi")
    return [(img, root / "annotations" / sample / (img.stem + ".json")) for img in images]
