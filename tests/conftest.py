import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    # /dev/kfd is the ROCm compute device node; avoids initialising HIP just to probe
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def orc_capi():
    from oracle import orc
    return orc.capi()


@pytest.fixture(scope="session")
def hip_capi():
    if not _have_gpu():
        pytest.skip("no GPU in this container")
    from renderer_amd.api import load_hip_capi
    return load_hip_capi()  # raises if the HIP library is not built: GPU tests must not pass without it
