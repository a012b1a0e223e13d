import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rust-raytracer_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def scenes_dir():
    return SCENES


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the C-ABI library and the oracle once per session (no-op when up to date)."""
    import __graft_entry__
    __graft_entry__.build()


def scene_path(name):
    return os.path.join(SCENES, name)


@pytest.fixture
def tuning():
    """rt_tuning_set for one test (the library reads no environment variable); defaults restored afterwards."""
    import rtamd
    yield rtamd.set_tuning
    rtamd.set_tuning()
