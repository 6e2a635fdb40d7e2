import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """CPU-side artefacts (oracle, workload generator) are built on demand; the
    HIP library is built by __graft_entry__.build() and travels prebuilt."""
    from oracle import oracle as O
    O.build()
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "workload")])
    yield
