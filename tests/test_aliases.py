"""Boundary: the seams stay importable under the reference's module paths.

`deephisto_amd.install_aliases()` registers them in sys.modules of a running process; the `compat/` tree
gives them to a fresh interpreter.  Every alias must BE the deephisto_amd object (nothing copied), and the
reference's own import lines (examples/predict_full_patched.py:12-19, models/patch_cls_simple/train.py:20-26)
must resolve."""
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]

REFERENCE_IMPORT_LINES = """
from anno.utils import AnnoDescription
from models.patch_cls_simple import utils
from models.patch_cls_simple.model import get_model
from patch_samplers.full_samplers import (
    FullImageDenseSampler,
    FullImageRndSampler,
    SamplerExecutionMode,
)
import models.patch_cls_simple.utils as utils2
from patch_samplers.region_samplers import AnnoRegionRndSampler
from utils import get_img_ano_paths
from examples.predict_full_patched import ImagePredictorPatched, batch_predictor, load_model
"""


def test_install_aliases_registers_the_same_objects():
    import deephisto_amd
    from deephisto_amd.aliases import ALIASES
    saved = {k: sys.modules.get(k) for k in ALIASES}
    try:
        for k in ALIASES:
            sys.modules.pop(k, None)
        names = deephisto_amd.install_aliases()
        assert set(names) == set(ALIASES)
        ns = {}
        exec(REFERENCE_IMPORT_LINES, ns)
        from deephisto_amd.examples import predict_full_patched as ours
        from deephisto_amd.models.patch_cls_simple import model as our_model
        from deephisto_amd.patch_samplers import full_samplers as our_fs
        assert ns["get_model"] is our_model.get_model
        assert ns["FullImageDenseSampler"] is our_fs.FullImageDenseSampler
        assert ns["SamplerExecutionMode"].INMEMORY_SINGLEPROC.value == 1 and ns["SamplerExecutionMode"].ONDISK_MULTIPROC.value == 2
        assert ns["ImagePredictorPatched"] is ours.ImagePredictorPatched and ns["batch_predictor"] is ours.batch_predictor
        # a name taken by somebody else is not silently replaced
        import types
        sys.modules["anno"] = types.ModuleType("anno")
        try:
            deephisto_amd.install_aliases()
            raise AssertionError("a foreign 'anno' module was overwritten")
        except RuntimeError:
            pass
        deephisto_amd.install_aliases(force=True)
        deephisto_amd.uninstall_aliases()
        assert "patch_samplers.full_samplers" not in sys.modules
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_compat_tree_serves_a_fresh_interpreter():
    env = {"PYTHONPATH": f"{REPO / 'compat'}:{REPO}", "PATH": "/usr/bin:/bin"}
    code = REFERENCE_IMPORT_LINES + """
import deephisto_amd.patch_samplers.full_samplers as fs
assert FullImageDenseSampler is fs.FullImageDenseSampler
print("ok")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd="/tmp")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
    # the reference's command line: python -m models.patch_cls_simple.train [--extract_test]
    r = subprocess.run([sys.executable, "-m", "models.patch_cls_simple.train", "--help"], capture_output=True, text=True,
                       env=env, cwd="/tmp")
    assert r.returncode == 0 and "--extract_test" in r.stdout, r.stderr[-2000:]
