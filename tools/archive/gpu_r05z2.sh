#!/bin/bash
# round 5, run z2: the quadrilateral tracer kernels with the data-carrying boundary facets after the outputs (140-168 VGPRs: three
# waves per SIMD) against the form with them inlined in the facet loop (176-198: two waves); tracer / quad / sipg / distributed tests
set -u
O=gpurun_out/r05z2; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_tracer.py tests/test_quads.py tests/test_gpu_sipg.py tests/test_gpu_fuzz.py tests/test_gpu_examples.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c1-300
for rep in 1 2; do
  for lib in new old; do
    if [ $lib = old ]; then export THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_tq_old.so; else unset THETIS_AMD_LIB; fi
    CFGBENCH_ONLY=quads timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | grep -i tracer | sed "s/^/$lib /" >> $O/quads_ab.txt
  done
done
unset THETIS_AMD_LIB
sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/quads_ab.txt | cut -c1-170
