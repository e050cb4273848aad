import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """Path of the built HIP library; builds it (hipcc cross-compiles without a GPU) if absent."""
    import medgp_amd
    if not os.path.exists(medgp_amd.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    return medgp_amd.lib_path()
