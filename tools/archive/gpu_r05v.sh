#!/bin/bash
# round 5, run v: the flow kernel stores a cell's step result only when it is the last one the launch computes for it ('product') against
# every step's result ('storeall', -DSWE_FLOW_STORE_EVERY_STEP); flow / distributed / spmd / fuzz / wetting-drying tests
set -u
O=gpurun_out/r05v; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_flow_kernel.py tests/test_distributed.py tests/test_gpu_spmd.py tests/test_gpu_fuzz.py tests/test_wetting_drying.py tests/test_gpu_solver2d.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c1-300
rb() { timeout 300 python tools/rankbench.py --case $1 --world 8 --rank $2 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1; }
kb() { THETIS_AMD_FLOW=1 timeout 300 python tools/kbench.py --nx $1 --ny $2 --steps 384 --prewarm 0.5 --tag flow1 2>&1 | tail -1; }
for rep in 1 2 3; do
  for v in product storeall; do
    if [ $v = product ]; then unset THETIS_AMD_LIB; else export THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so; fi
    rb cfg2 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb cfg2 0 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb cfg5 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    kb 354 177 | sed "s/^/$v /" >> $O/flow_ab.txt
    kb 250 125 | sed "s/^/$v /" >> $O/flow_ab.txt
  done
done
unset THETIS_AMD_LIB
sed 's/"exchange.*"rank"/ "rank"/; s/"every.*"us_per_step"/ us_per_step/' $O/rank_ab.txt
sed 's/"order.*"n_cells"/"n_cells"/; s/, "us_per_launch.*//' $O/flow_ab.txt
