#!/bin/bash
# round 5, run w: received ghost values go to the state planes in the last cycle of a launch only; in-launch-exchange tests + rank rows
set -u
O=gpurun_out/r05w; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_distributed.py tests/test_gpu_spmd.py tests/test_gpu_flow_kernel.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c1-300
rb() { timeout 300 python tools/rankbench.py --case $1 --world 8 --rank $2 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1; }
for rep in 1 2 3; do rb cfg2 3 >> $O/rank.txt; rb cfg2 0 >> $O/rank.txt; done
sed 's/"exchange.*"rank"/ "rank"/; s/"every.*"us_per_step"/ us_per_step/' $O/rank.txt
timeout 600 python tools/kbench.py --steps 100 --prewarm 0.5 --tag cfg2_1M 2>&1 | tail -1 | cut -c1-200
