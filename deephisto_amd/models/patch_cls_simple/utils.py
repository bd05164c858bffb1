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
