"""Builds thetis_amd/libswe2d_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU.

The library is several translation units (csrc/swe2d_api*.hip: the C ABI by concern; csrc/swe2d_k_*.hip: one kernel family
each - the template instantiations are what takes the time).  Objects are compiled in parallel and only when a file they include
changed (hipcc -MD dependency files), so a kernel edit rebuilds the families that see it and nothing else.

``build(defines=[...], lib=...)`` builds a variant (its objects in a directory of its own next to ``lib``): the adversaries of the
granule protocol, -DSWE_FLOW_DELAY / -DSWE_FLOW_TEAR, whose device-side switches live in the flow kernels' unit.
``build(unity=True, defines=[...], lib=...)`` compiles csrc/swe2d_unity.hip instead - every unit in one, for the debug variants
whose device-side globals all kernels must share (-DSWE_RANGE_CHECK, -DSWE_WAVE_TIMING)."""
import concurrent.futures
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
UNITS = ['swe2d_api.hip', 'swe2d_api_flow.hip', 'swe2d_api_tracer.hip', 'swe2d_api_p2p.hip', 'swe2d_api_fuse.hip',
         'swe2d_k_tri.hip', 'swe2d_k_wd.hip', 'swe2d_k_quad.hip', 'swe2d_k_flow.hip', 'swe2d_k_flow_wd.hip', 'swe2d_k_tracer.hip']
UNITY = os.path.join(CSRC, 'swe2d_unity.hip')
OBJ_DIR = os.path.join(CSRC, '.obj')
LIB = os.path.join(_HERE, 'libswe2d_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']
# Per-unit compiler flags.  The dataflow kernels and the quadrilateral kernels are compiled WITHOUT machine LICM: hoisted literal
# materialisations and address arithmetic stay live across the stage loop / the quadrature, which cost the flow kernels 14-20 VGPRs
# and every variant with source terms its scratch (254-256 VGPRs + 12-28 B/lane -> 224-242, none), the quadrilateral kernels the
# third wave of their remaining instances (profiles/r05m_no_machine_licm.txt: flow kernel on one device 17.1 -> 16.8 us per step,
# quadrilaterals -1.5 ... -2 %).  Not for the triangle stage kernels (no change, although the first-stage epilogue variant reaches
# four waves per SIMD) and not for the wetting-drying unit (cfg 5 +3 %).
_NO_MLICM = ['-mllvm', '-disable-machine-licm']
UNIT_FLAGS = {'swe2d_api_fuse.hip': _NO_MLICM, 'swe2d_k_flow.hip': _NO_MLICM, 'swe2d_k_flow_wd.hip': _NO_MLICM, 'swe2d_k_quad.hip': _NO_MLICM}
# every file a translation unit can include (the fallback when an object has no dependency file yet)
ALL_DEPS = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(_HERE, '..', 'include', 'swe2d.h')]


def _obj(unit, obj_dir=None):
    return os.path.join(obj_dir or OBJ_DIR, unit[:-4] + '.o')


def _deps(unit):
    """Files the object of `unit` was built from, by the compiler's own dependency file."""
    d = _obj(unit)[:-2] + '.d'
    if not os.path.exists(d):
        return None
    text = open(d).read().replace('\\\n', ' ')
    out = []
    for part in text.split(':', 1)[1].split():
        if part.startswith('/opt/rocm') or part.startswith('/usr/'):
            continue
        out.append(part)
    return out


def _stale(unit):
    o = _obj(unit)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    deps = _deps(unit)
    if deps is None:
        deps = [os.path.join(CSRC, unit)] + ALL_DEPS
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if any(os.path.getmtime(os.path.join(CSRC, u)) > t for u in UNITS) or any(os.path.getmtime(d) > t for d in ALL_DEPS):
        return True
    return False


def _compile(unit, verbose, defines=(), obj_dir=None, extra_flags=()):
    o = _obj(unit, obj_dir)
    cmd = [HIPCC] + FLAGS + list(UNIT_FLAGS.get(unit, [])) + list(extra_flags) + ['-D' + d for d in defines] \
        + ['-c', os.path.join(CSRC, unit), '-o', o, '-MD', '-MF', o[:-2] + '.d']
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose=False, unity=False, defines=(), lib=None, jobs=None, extra_flags=()):
    """Compile the HIP extension; returns the path of the shared library.  ``extra_flags``: compiler flags of a variant build (A/B)."""
    if unity:
        out = lib or LIB
        cmd = [HIPCC] + FLAGS + list(extra_flags) + ['-shared'] + ['-D' + d for d in defines] + [UNITY, '-o', out]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        return out
    jobs = jobs or int(os.environ.get('THETIS_AMD_BUILD_JOBS', '0')) or min(len(UNITS), os.cpu_count() or 1)
    if defines or lib or extra_flags:
        # a variant: all units with the extra defines, objects next to the variant's library
        out = os.path.abspath(lib or LIB)
        obj_dir = out + '.obj'
        os.makedirs(obj_dir, exist_ok=True)
        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
            for f in [ex.submit(_compile, u, verbose, defines, obj_dir, extra_flags) for u in UNITS]:
                f.result()
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + [_obj(u, obj_dir) for u in UNITS] + ['-o', out])
        return out
    if not (force or needs_build()):
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = [u for u in UNITS if force or _stale(u)]
    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
            for f in [ex.submit(_compile, u, verbose) for u in todo]:
                f.result()
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + [_obj(u) for u in UNITS] + ['-o', LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    import sys
    print(build(force='--force' in sys.argv, verbose=True))
