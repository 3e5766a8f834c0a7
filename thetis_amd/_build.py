"""Builds thetis_amd/libswe2d_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, 'csrc', 'swe2d_api.hip')
import glob
# every header the translation unit can include: editing any of them must trigger a rebuild
DEPS = [SRC] + sorted(glob.glob(os.path.join(_HERE, 'csrc', '*.h'))) + [os.path.join(_HERE, '..', 'include', 'swe2d.h')]
LIB = os.path.join(_HERE, 'libswe2d_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared']


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    """Compile the HIP extension; returns the path of the shared library."""
    if force or needs_build():
        cmd = [HIPCC] + FLAGS + [SRC, '-o', LIB]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
