"""deephisto_amd -- MI355X-native whole-slide patch pipeline.

Drop-in for the deephisto hot path (patch_samplers.full_samplers.FullImageDenseSampler
-> examples.predict_full_patched -> models.patch_cls_simple): Python host code
mirrors the reference's interface; all computation runs in hand-written HIP
kernels for gfx950 behind the C ABI of include/deephisto_hip.h.
"""
__version__ = "0.1.0"

from .aliases import install_aliases, uninstall_aliases  # noqa: E402,F401  (imports nothing heavy: modules load on call)
