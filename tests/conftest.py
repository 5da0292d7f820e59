import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Builds the oracle and (cross-compiles) the HIP library once per session."""
    from defuse_amd import build
    build.build_oracle()
    build.build_lib()
    return True


@pytest.fixture(scope="session")
def gpu_ctx(built):
    from defuse_amd import dsa
    ctx = dsa.Context(0)   # raises when there is no GPU or no library: GPU tests must not fall back
    yield ctx
    ctx.close()
