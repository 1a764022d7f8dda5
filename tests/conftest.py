import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import diffusiontexturepainting_amd  # noqa: E402,F401  (with $DTP_RUNTIME_ENV=1: the optional HIP runtime configuration, before any test imports torch)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "experimental: exercises a measured-and-switched-off experiment that only a DTP_EXPERIMENTAL=1 build of "
                                       "libdtp.so contains (gemmws_kernel, GroupNorm on the halo conv's staged patch); skipped on the default build")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
