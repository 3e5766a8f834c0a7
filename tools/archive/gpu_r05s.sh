#!/bin/bash
# round 5, run s: the flow kernel's blocks as 8 x 4-quad tiles (ordering.flow_block_order: 24-31 rim facets per block, four granule
# loads per lane and polling pass) against the two-row blocks of the device numbering (THETIS_AMD_FLOW_BLOCKS=0: 36-68 rim facets,
# eight / nine loads); flow / distributed / spmd tests
set -u
O=gpurun_out/r05s; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_flow_kernel.py tests/test_distributed.py tests/test_gpu_spmd.py tests/test_gpu_fuzz.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c1-300
rb() { timeout 300 python tools/rankbench.py --case $1 --world 8 --rank $2 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1; }
kb() { THETIS_AMD_FLOW=1 timeout 300 python tools/kbench.py --nx $1 --ny $2 --steps 384 --prewarm 0.5 --tag flow1 2>&1 | tail -1; }
for rep in 1 2; do
  for v in blocks rows; do
    if [ $v = rows ]; then export THETIS_AMD_FLOW_BLOCKS=0; else unset THETIS_AMD_FLOW_BLOCKS; fi
    rb cfg2 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb cfg2 0 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb cfg2_src 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb cfg5 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    kb 354 177 | sed "s/^/$v /" >> $O/flow_ab.txt
    kb 250 125 | sed "s/^/$v /" >> $O/flow_ab.txt
  done
done
unset THETIS_AMD_FLOW_BLOCKS
sed 's/"exchange.*"rank"/ "rank"/; s/"every.*"us_per_step"/ us_per_step/' $O/rank_ab.txt
sed 's/"order.*"n_cells"/"n_cells"/; s/, "us_per_launch.*//' $O/flow_ab.txt
