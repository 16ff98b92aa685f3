"""pytest configuration: markers, library fixtures.  GPU tests are marked `gpu`; everything else runs on CPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle.  Built on demand here (gcc only); on the GPU box the prebuilt .so travels with the tree."""
    import hx
    if not os.path.exists(hx.ORACLE_SO):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True, capture_output=True)
    return hx.oracle()


@pytest.fixture(scope="session")
def rtc():
    import raytracer_amd as ra
    ra.lib()
    return ra


@pytest.fixture(scope="session")
def gpu(rtc):
    rtc.rtc_init()
    return rtc
