"""Reference path models/patch_cls_simple/utils.py."""
from deephisto_amd.models.patch_cls_simple.utils import get_device, get_img_ano_paths, load_config  # noqa: F401
