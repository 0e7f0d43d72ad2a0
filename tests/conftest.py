import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_port():
    from oracle import port
    port.lib()
    return port


@pytest.fixture(scope="session")
def oracle_ref():
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    try:
        ref.lib()
    except OSError as e:  # e.g. LAPACK provider missing on this box
        pytest.skip(f"oracle/_ref not loadable: {e}")
    return ref
