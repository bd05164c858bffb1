import json
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_meta():
    return json.loads((GOLDEN / "golden.json").read_text())


@pytest.fixture(scope="session")
def golden_grids():
    return dict(np.load(GOLDEN / "grids.npz"))


@pytest.fixture(scope="session")
def golden_vectors():
    return dict(np.load(GOLDEN / "vectors.npz"))


@pytest.fixture(scope="session")
def built_lib():
    """The in-tree C-ABI library; built here if the snapshot does not carry it."""
    from deephisto_amd import _lib, build
    if not _lib.LIB_PATH.exists():
        build.build_library(verbose=False)
    return _lib.lib()
