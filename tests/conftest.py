import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def rt():
    """The product binding; GPU tests fail loudly if librtmi.so is missing or no device."""
    import raytracer_amd
    n = raytracer_amd.lib().rtmi_device_count()
    assert n >= 1, f"no gfx950 device visible: {raytracer_amd.lib().rtmi_last_error()}"
    return raytracer_amd
