import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def blue_noise():
    """The reference's 512x512 RGBA8 blue-noise table (src/render/pipeline/blue_noise_512.png decoded once;
    sha256 of the raw bytes is pinned in tests/test_fixtures.py)."""
    return np.fromfile(os.path.join(GOLDEN, "blue_noise_512.rgba"), dtype=np.uint8)


@pytest.fixture(scope="session")
def native_built():
    from raytrace_amd import build
    build.build()
    return True


@pytest.fixture(scope="session")
def procedural_region(native_built):
    from raytrace_amd import world
    return world.generate_region(world.DEFAULT_SEED)
