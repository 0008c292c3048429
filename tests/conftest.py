import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(GOLDEN / f"{name}.npz", allow_pickle=False)

    return load


@pytest.fixture(scope="session")
def hip_lib():
    """libpram_hip.so built in-tree (built on demand when hipcc is available)."""
    from pram_amd import _lib, build
    if not _lib.lib_path().exists():
        build.build(verbose=False)
    return _lib.load()
