import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """Builds (if needed) and loads the CPU oracle."""
    from oracle import binding

    binding.build()
    return binding.load()


@pytest.fixture(scope="session")
def renderer():
    """One HIP renderer context for the whole GPU session (one process, one context)."""
    import volumetricraytracer_amd as v

    r = v.VHipRenderer()
    if not r.Start():
        pytest.fail("VHipRenderer.Start() failed on a GPU-marked test: no device or library error")
    yield r
    r.Stop()
