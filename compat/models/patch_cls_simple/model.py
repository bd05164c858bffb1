"""Reference path models/patch_cls_simple/model.py."""
from deephisto_amd.models.patch_cls_simple.model import get_model  # noqa: F401
