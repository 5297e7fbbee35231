import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU oracle run (excluded from the default CPU suite)")


@pytest.fixture(scope="session")
def oracle_built():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def eng():
    """The HIP engine binding; fails loudly (no CPU fallback) when there is no gfx950."""
    import fluid_sims_amd as f
    L = f.load()
    assert L.tau_device_available() == 1, "gpu-marked test needs a gfx950 device"
    return f
