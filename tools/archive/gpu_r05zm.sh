#!/bin/bash
# round 5, run zm: the fused stage pair with one trace address pair per facet (168 VGPRs WITHOUT scratch) against the build with three
# (28 B/lane of scratch: build_dbg/libswe2d_fuse_tr3.so); parity tests
set -u
O=gpurun_out/r05zm; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_contract.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log | cut -c1-200
for rep in 1 2 3; do
  for v in product tr3; do
    if [ $v = product ]; then unset THETIS_AMD_LIB; else export THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_fuse_$v.so; fi
    for sz in "707 354" "1000 500" "2000 1000"; do
      set -- $sz
      THETIS_AMD_FLOW=0 timeout 300 python tools/kbench.py --nx $1 --ny $2 --steps 60 --prewarm 0.5 2>&1 | grep "^{" | sed "s/^/$v $1x$2 /" >> $O/kbench_ab.txt
    done
  done
done
unset THETIS_AMD_LIB
cut -c1-150 $O/kbench_ab.txt
