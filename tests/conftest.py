import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionfinish(session, exitstatus):
    """Against the -DSWE_RANGE_CHECK build (tools/range_check.sh, THETIS_AMD_LIB=...): fail the session when a kernel
    touched memory outside the library's own allocations."""
    if not os.environ.get('THETIS_AMD_LIB'):
        return
    import ctypes
    from thetis_amd import _lib
    lib = _lib.load()
    if not hasattr(lib, 'swe2d_debug_range_report'):
        return
    out = (ctypes.c_ulonglong*5)()
    lib.swe2d_debug_range_report.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    lib.swe2d_debug_range_report(out)
    print('\nrange check: {:d} checked launches, {:d} violations (first: address 0x{:x}, line {:d})'.format(out[3], out[0], out[1], out[2]))
    if out[0]:
        session.exitstatus = 1


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
