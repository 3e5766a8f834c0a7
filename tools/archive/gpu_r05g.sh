#!/bin/bash
# round 5, run g: the quadrilateral stage kernel with its boundary facets after the outputs and the optional terms in a pass of their own
# (three waves per SIMD) against the round-4 form (build_dbg/libswe2d_quad_old.so, -DSWE_QUAD_OLD), same box; quadrilateral tests
set -u
O=gpurun_out/r05g; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_quads.py tests/test_gpu_parity.py tests/test_gpu_tracer.py tests/test_wetting_drying.py tests/test_gpu_fuzz.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gpu_tests.log | cut -c1-300
for rep in 1 2; do
  for lib in new old; do
    if [ $lib = old ]; then export THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_quad_old.so; else unset THETIS_AMD_LIB; fi
    CFGBENCH_ONLY=quads timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/$lib /" >> $O/quads_ab.txt
  done
done
unset THETIS_AMD_LIB
cut -c1-160 $O/quads_ab.txt
