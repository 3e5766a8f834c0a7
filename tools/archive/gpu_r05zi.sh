#!/bin/bash
# round 5, run zi: the scalars of the end of a stage (beta, whether and where to publish) held in vector registers across the wait (pinargs, -DSWE_FLOW_PIN_ARGS) against the product
set -u
O=gpurun_out/r05zi; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_pinargs.so timeout 1200 python -m pytest tests/test_gpu_flow_kernel.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log | cut -c1-300
rb() { timeout 300 python tools/rankbench.py --case $1 --world 8 --rank $2 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1; }
for rep in 1 2 3; do
  for v in product pinargs; do
    if [ $v = product ]; then unset THETIS_AMD_LIB; else export THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so; fi
    rb cfg2 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb cfg2 0 | sed "s/^/$v /" >> $O/rank_ab.txt
  done
done
unset THETIS_AMD_LIB
sed 's/"exchange.*"rank"/ "rank"/; s/"every.*"us_per_step"/ us_per_step/' $O/rank_ab.txt
