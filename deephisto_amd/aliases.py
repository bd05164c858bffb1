"""Reference module paths for the drop-in seams.

The reference's callers import the hot path as
    patch_samplers.full_samplers / patch_samplers.region_samplers
    models.patch_cls_simple.{model, train, utils}
    examples.predict_full_patched
    anno.utils
    utils                                  (top-level: get_img_ano_paths)
(`examples/predict_full_patched.py:12-19`, `models/patch_cls_simple/train.py:20-26` of the reference).
`install_aliases()` registers this package's modules under those names in `sys.modules`, so such a
caller runs unchanged inside a process that has called it; the `compat/` directory at the repository
root gives the same names to a fresh interpreter (`PYTHONPATH=compat python -m models.patch_cls_simple.train`).
Nothing is copied: every alias IS the `deephisto_amd` module.
"""
from __future__ import annotations

import importlib
import sys

#: reference module path -> module of this package
ALIASES = {
    "patch_samplers": "deephisto_amd.patch_samplers",
    "patch_samplers.full_samplers": "deephisto_amd.patch_samplers.full_samplers",
    "patch_samplers.region_samplers": "deephisto_amd.patch_samplers.region_samplers",
    "models": "deephisto_amd.models",
    "models.patch_cls_simple": "deephisto_amd.models.patch_cls_simple",
    "models.patch_cls_simple.model": "deephisto_amd.models.patch_cls_simple.model",
    "models.patch_cls_simple.train": "deephisto_amd.models.patch_cls_simple.train",
    "models.patch_cls_simple.utils": "deephisto_amd.models.patch_cls_simple.utils",
    "examples": "deephisto_amd.examples",
    "examples.predict_full_patched": "deephisto_amd.examples.predict_full_patched",
    "examples.sample_full_dense": "deephisto_amd.examples.sample_full_dense",
    "anno": "deephisto_amd.anno",
    "anno.utils": "deephisto_amd.anno.utils",
    "utils": "deephisto_amd.models.patch_cls_simple.utils",
}


def install_aliases(force: bool = False) -> list[str]:
    """Register the reference's module paths.  A name that is already taken by a DIFFERENT module is left
    alone and reported by a RuntimeError unless `force` (generic names such as `models` or `utils` may belong
    to the host application).  Returns the names registered."""
    done = []
    for name, target in ALIASES.items():
        mod = importlib.import_module(target)
        cur = sys.modules.get(name)
        if cur is not None and cur is not mod and not force:
            raise RuntimeError(f"module name '{name}' is already taken by {getattr(cur, '__file__', cur)!r}; "
                               "call install_aliases(force=True) to replace it")
        sys.modules[name] = mod
        done.append(name)
    # attribute access on the parents (`import patch_samplers.full_samplers as fs` resolves through sys.modules,
    # `patch_samplers.full_samplers` as an attribute needs the parent to carry it -- the real packages already do)
    return done


def uninstall_aliases() -> None:
    for name, target in ALIASES.items():
        if name in sys.modules and sys.modules[name] is sys.modules.get(target):
            del sys.modules[name]
