"""Reference path anno/utils.py."""
from deephisto_amd.anno.utils import AnnoClass, AnnoDescription  # noqa: F401
