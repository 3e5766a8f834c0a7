#!/bin/bash
# round 5, run zf: the flow kernel's cell integrals PINNED in front of the polling loop ('pin': -DSWE_FLOW_PIN_CELL_TERMS; the compiler had
# sunk them behind it), and the Shu-Osher weights as well ('pinw': + -DSWE_FLOW_PIN_W), against the product; bitwise tests with 'pinw'
set -u
O=gpurun_out/r05zf; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_pinw.so timeout 2400 python -m pytest tests/test_gpu_flow_kernel.py tests/test_distributed.py tests/test_gpu_spmd.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c1-300
rb() { timeout 300 python tools/rankbench.py --case $1 --world 8 --rank $2 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1; }
kb() { THETIS_AMD_FLOW=1 timeout 300 python tools/kbench.py --nx $1 --ny $2 --steps 384 --prewarm 0.5 --tag flow1 2>&1 | tail -1; }
for rep in 1 2 3; do
  for v in product pin pinw; do
    if [ $v = product ]; then unset THETIS_AMD_LIB; else export THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so; fi
    rb cfg2 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb cfg2 0 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb cfg5 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb cfg2_src 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    kb 354 177 | sed "s/^/$v /" >> $O/flow_ab.txt
  done
done
unset THETIS_AMD_LIB
sed 's/"exchange.*"rank"/ "rank"/; s/"every.*"us_per_step"/ us_per_step/' $O/rank_ab.txt
sed 's/"order.*"n_cells"/"n_cells"/; s/, "us_per_launch.*//' $O/flow_ab.txt
