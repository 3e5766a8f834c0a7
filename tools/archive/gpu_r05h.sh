#!/bin/bash
# round 5, run h: the polling pass of the flow kernel - granules that have arrived are not read again (got), padding granules are
# neither stored nor polled (nopads) - against the form of run f; rank 3 of 8 and one device at 125 k cells; flow kernel tests
set -u
O=gpurun_out/r05h; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_flow_kernel.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c1-300
rb() { timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1; }
kb() { THETIS_AMD_FLOW=1 timeout 300 python tools/kbench.py --nx 354 --ny 177 --steps 384 --prewarm 0.5 --tag flow1 2>&1 | tail -1; }
for rep in 1 2; do
  for v in product nogot oldpoll5; do
    if [ $v = product ]; then rb | sed "s/^/$v /" >> $O/rank_ab.txt; kb | sed "s/^/$v /" >> $O/flow_ab.txt
    else THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so rb | sed "s/^/$v /" >> $O/rank_ab.txt; THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so kb | sed "s/^/$v /" >> $O/flow_ab.txt; fi
  done
done
sed 's/{.*"us_per_step"/ us_per_step/' $O/rank_ab.txt
sed 's/{.*"us_per_step"/ us_per_step/; s/, "us_per_launch.*//' $O/flow_ab.txt
