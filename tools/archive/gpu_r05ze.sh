#!/bin/bash
# round 5, run ze: quadrilateral kernels read the connectivity as 24-B records (csrc/swe2d_conn.h) instead of eight planes -
# parity tests, then the quadrilateral rows with the records against THETIS_AMD_COMPACT_IDX=0, alternating
set -u
O=gpurun_out/r05ze; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_quads.py tests/test_gpu_fuzz.py tests/test_gpu_tracer.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c1-300
for rep in 1 2 3; do
  for c in 1 0; do
    export THETIS_AMD_COMPACT_IDX=$c
    CFGBENCH_ONLY=quads timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/compact=$c /" >> $O/quads_ab.txt
  done
done
for c in 1 0; do
  export THETIS_AMD_COMPACT_IDX=$c
  CFGBENCH_QUAD_N=800 CFGBENCH_ONLY=quads timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/compact=$c 640k /" >> $O/quads_ab.txt
done
unset THETIS_AMD_COMPACT_IDX
sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/quads_ab.txt | cut -c1-190
