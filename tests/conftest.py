import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def renderer():
    """One rt_ctx on cuda:0 for the whole GPU session (tests re-set scene / size as needed)."""
    import raytracing_engine_amd as R

    r = R.Renderer(0)  # raises (no fallback) when librt_amd.so or the GPU is missing
    yield r
    r.close()
