"""Reference top-level utils.py (get_img_ano_paths, train.py:26)."""
from deephisto_amd.models.patch_cls_simple.utils import get_img_ano_paths  # noqa: F401
