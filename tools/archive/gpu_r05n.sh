#!/bin/bash
# round 5, run n: the flow kernel compiled without machine LICM + polling offsets computed once per stage + the wall fast path in the
# source-term variants ('product') against the same without the hoisted offsets ('nohoist') and against the library of run i
# ('r05i': LICM on, offsets per pass, general wall path in the source-term variants); flow / distributed tests on the product
set -u
O=gpurun_out/r05n; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_flow_kernel.py tests/test_distributed.py tests/test_gpu_fuzz.py tests/test_quads.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c1-300
rb() { timeout 300 python tools/rankbench.py --case $1 --world 8 --rank $2 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1; }
kb() { THETIS_AMD_FLOW=1 timeout 300 python tools/kbench.py --nx 354 --ny 177 --steps 384 --prewarm 0.5 --tag flow1 2>&1 | tail -1; }
for rep in 1 2; do
  for v in product nohoist r05i; do
    if [ $v = product ]; then unset THETIS_AMD_LIB; else export THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so; fi
    rb cfg2 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb cfg2 0 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb cfg2_src 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    kb | sed "s/^/$v /" >> $O/flow_ab.txt
  done
done
unset THETIS_AMD_LIB
sed 's/"exchange.*"rank"/ "rank"/; s/"every.*"us_per_step"/ us_per_step/' $O/rank_ab.txt
sed 's/"order.*"n_cells"/"n_cells"/; s/, "us_per_launch.*//' $O/flow_ab.txt
