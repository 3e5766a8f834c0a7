#!/bin/bash
# round 5, run u: the adversary builds of the granule protocol with the six-granule polling pass and the tile blocks; wetting-drying
# instance with the polling offsets computed once per stage (three registers) against once per pass
set -u
O=gpurun_out/r05u; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_delay.so timeout 1200 python -m pytest tests/test_gpu_flow_kernel.py tests/test_distributed.py -m gpu -q -k "lags or lagging" 2>&1 | tail -3 | sed "s/^/[delay] /" | tee -a $O/adversaries.txt
for v in tear tear_nocheck; do
  THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so timeout 1200 python -m pytest tests/test_gpu_flow_kernel.py tests/test_distributed.py tests/test_gpu_spmd.py -m gpu -q -k "two_halves or torn or periodic_verification" 2>&1 | tail -3 | sed "s/^/[$v] /" | tee -a $O/adversaries.txt
done
timeout 900 python -m pytest tests/test_gpu_flow_kernel.py tests/test_wetting_drying.py -q -m gpu 2>&1 | tail -2
rb() { timeout 300 python tools/rankbench.py --case $1 --world 8 --rank $2 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1; }
for rep in 1 2 3; do
  for v in product wdnohoist; do
    if [ $v = product ]; then unset THETIS_AMD_LIB; else export THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so; fi
    rb cfg5 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb cfg5 7 | sed "s/^/$v /" >> $O/rank_ab.txt
  done
done
unset THETIS_AMD_LIB
sed 's/"exchange.*"rank"/ "rank"/; s/"every.*"us_per_step"/ us_per_step/' $O/rank_ab.txt
