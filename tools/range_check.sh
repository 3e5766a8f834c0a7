#!/bin/bash
# Range-check pass over the device code (the image has no ASAN-enabled HIP runtime).
# Builds the library with -DSWE_RANGE_CHECK (swe2d_kernels.h: every raw-buffer access tested against the allocation table) and
# runs tools/range_check.py, its negative control, and the GPU parity + fuzz tests against it (tests/conftest.py reads the
# report at session end and fails the session on a violation).
#   bash tools/range_check.sh [build]      (build: compile build_dbg/libswe2d_rangecheck.so first; hipcc cross-compiles without a GPU)
set -u
cd "$(dirname "$0")/.."
if [ "${1:-}" = build ] || [ ! -f build_dbg/libswe2d_rangecheck.so ]; then
  mkdir -p build_dbg
  python - <<'PY' || exit 1
from thetis_amd import _build
_build.build(unity=True, defines=['SWE_RANGE_CHECK'], lib='build_dbg/libswe2d_rangecheck.so')
_build.build(defines=['SWE_FLOW_DELAY'], lib='build_dbg/libswe2d_delay.so')
_build.build(defines=['SWE_FLOW_TEAR'], lib='build_dbg/libswe2d_tear.so')
_build.build(defines=['SWE_FLOW_TEAR', 'SWE_FLOW_NOCHECK'], lib='build_dbg/libswe2d_tear_nocheck.so')
PY
  [ "${1:-}" = build ] && exit 0
fi
export THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_rangecheck.so
timeout 900 python tools/range_check.py; echo "range_check rc=$?"
THETIS_AMD_RANGE_SELFTEST=1 timeout 900 python tools/range_check.py | tail -1; echo "negative control rc=$?"
# the adversary of the granule protocol: the same library with -DSWE_FLOW_DELAY (no range checks: the timing is the point)
THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_delay.so timeout 1200 python -m pytest tests/test_gpu_flow_kernel.py tests/test_distributed.py -m gpu -q \
    -k "lags or lagging" 2>&1 | tail -5
# the adversary of the granules' check word: stores made in two halves (-DSWE_FLOW_TEAR); then the negative control - the same
# build with consumers that look at the tag only (-DSWE_FLOW_NOCHECK) must give wrong bits, which the tests assert
for v in tear tear_nocheck; do
  THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so timeout 1200 python -m pytest tests/test_gpu_flow_kernel.py tests/test_distributed.py tests/test_gpu_spmd.py -m gpu -q \
      -k "two_halves or torn or periodic_verification" 2>&1 | tail -3 | sed "s/^/[$v] /"
done
# the checked kernels are an order of magnitude slower: the small-mesh tests only
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_tracer.py tests/test_gpu_sipg.py tests/test_quads.py tests/test_gpu_flow_kernel.py \
    -m gpu -q -k "not large_launch and not full_size" 2>&1 | tail -60
