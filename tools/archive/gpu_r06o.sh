#!/bin/bash
# round 6, run 15: step launches inside coupled cycles - tests, then the cfg 4 ranks with and without
set -u
TAG=r06o
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest -q -m gpu tests/test_distributed.py -k "whole_steps_in_one_launch or coupled" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/tests.log | tail -20 | cut -c1-250
rb() { timeout 400 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/${TAG}_rank.txt; }
for f in 2 auto 2 auto; do
  if [ $f = auto ]; then unset THETIS_AMD_FUSE12; else export THETIS_AMD_FUSE12=$f; fi
  echo "--- THETIS_AMD_FUSE12=$f" >> $O/${TAG}_rank.txt
  rb --case cfg4 --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480
  rb --case cfg4 --world 8 --rank 3 --every 4 --exchange p2p --graph-mode full --steps 480
  rb --case cfg4 --world 2 --rank 0 --every 2 --exchange p2p --graph-mode full --steps 480
done
unset THETIS_AMD_FUSE12
sed 's/"exchange.*"rank"/ "rank"/; s/"every":/ every/; s/"overlap.*"fused_pair"/ "fused_pair"/' $O/${TAG}_rank.txt | cut -c1-200
