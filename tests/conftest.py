import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Creates device contexts; fails loudly (no skip, no fallback) if the HIP extension or the GPU is missing."""
    from nexus_amd import capi

    assert os.path.exists(capi.LIB_PATH), "libnexus_amd.so has not been built"
    assert capi.device_count() >= 1, "no HIP device visible: -m gpu tests must run on the GPU box"
    made = []

    def make(width, height):
        c = capi.Context(width, height)
        made.append(c)
        return c

    yield make
    for c in made:
        c.close()
