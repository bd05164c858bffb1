"""Mirror of models/patch_cls_simple/utils.py:1-17 (config loader, device picker)."""


def load_config(config_path):
    import yaml

    with open(config_path, "r") as file:
        return yaml.safe_load(file)


def get_device():
    """Same preference order as the reference (utils.py:8-17): mps if built, else
    cuda, else cpu.  On PyTorch-ROCm `mps.is_built()` is False and an MI355X shows
    up as "cuda"."""
    import torch

    if torch.backends.mps.is_built():
        device = torch.device("mps")
    elif torch.cuda.is_available():
        device = torch.device("cuda")
    else:
        device = torch.device("cpu")
    return device


def get_img_ano_paths(ds_folder, sample: str = "train"):
    """(image, annotation) path pairs of a dataset folder: `images/<sample>/*.psi` with
    `annotations/<sample>/<stem>.json` (the reference's top-level utils.py:4-14)."""
    from pathlib import Path

    ds_folder = Path(ds_folder)
    img_paths = [p for p in (ds_folder / "images" / sample).iterdir() if p.is_file() and p.suffix == ".psi"]
    return [(p, ds_folder / "annotations" / sample / f"{p.stem}.json") for p in img_paths]
