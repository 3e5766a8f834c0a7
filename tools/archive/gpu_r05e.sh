#!/bin/bash
# round 5, fifth run: several blocks per wave (csrc/swe2d_mflow.h) - bits against the stage launches, ranks of four and two of the bench
# mesh and one device at 250 k / 500 k / 1 M cells with stage launches and with the multi-block flow launches
set -u
O=gpurun_out/r05e; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_flow_kernel.py -q -m gpu -x > $O/flow_tests.log 2>&1; echo "flow tests rc=$?"; tail -12 $O/flow_tests.log | cut -c1-300
rb() { timeout 300 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/rank.txt; }
for fl in 0 1; do
  rb --world 4 --rank 1 --every 4 --exchange p2p --nosplit --flow $fl --flowx 0 --graph-mode full --steps 960
  rb --world 2 --rank 0 --every 4 --exchange p2p --nosplit --flow $fl --flowx 0 --graph-mode full --steps 960
done
rb --world 4 --rank 1 --every 8 --exchange p2p --nosplit --flow 1 --flowx 0 --graph-mode full --steps 960
rb --world 4 --rank 1 --every 2 --exchange p2p --nosplit --flow 1 --flowx 0 --graph-mode full --steps 960
sed 's/{.*"world"/ world/; s/"overlap.*"flow"/ flow/' $O/rank.txt
for nx in 500 708 1000; do for fl in 0 1; do
  THETIS_AMD_MFLOW_ADVANCE_K=8 THETIS_AMD_FLOW=$fl timeout 300 python tools/kbench.py --nx $nx --ny $((nx/2)) --steps 96 --prewarm 0.5 --tag flow$fl 2>&1 | tail -1 >> $O/flow_sizes.txt
done; done
cut -c1-200 $O/flow_sizes.txt
