import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def hip_lib():
    """The HIP extension; built in-tree if missing (hipcc cross-compiles on CPU)."""
    from thetis_amd import _build, _lib
    if _build.needs_build():
        _build.build()
    return _lib.load()


@pytest.fixture(scope='session')
def ref_so():
    from oracle import ref_lib
    ref_lib.build()
    return ref_lib
