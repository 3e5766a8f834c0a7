#!/bin/bash
# round 5, run z4: triangle tracer kernels - the epilogue form in the instances with the diffusion fused in only ('new'), against
# the library before the change ('old'); tracer / sipg / fuzz / distributed tests
set -u
O=gpurun_out/r05z4; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_tracer.py tests/test_gpu_sipg.py tests/test_gpu_fuzz.py tests/test_gpu_examples.py tests/test_distributed.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c1-300
for rep in 1 2; do
  for lib in new old; do
    if [ $lib = new ]; then unset THETIS_AMD_LIB; else export THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_tt_$lib.so; fi
    CFGBENCH_ONLY=tracers timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/$lib /" >> $O/tri_ab.txt
  done
done
unset THETIS_AMD_LIB
sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/tri_ab.txt | cut -c1-170
