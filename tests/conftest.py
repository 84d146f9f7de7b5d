import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "default_policy: LU tests that run under the default policy (tests/test_gpu_lu.py)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build(ref=os.path.isdir("/root/reference"))
    return pyoracle.Oracle()


@pytest.fixture(scope="session")
def ref(oracle):
    from oracle import pyoracle
    if not pyoracle.ref_available():
        pytest.skip("oracle/_ref not built (reference sources absent)")
    return pyoracle.Ref()
