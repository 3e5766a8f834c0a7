#!/bin/bash
# round 6, run 14: a partition's steps in one launch each (swe2d_solve_step_cells) - bitwise tests, then ranks of two / four / eight with and without
set -u
TAG=r06n
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest -q -m gpu tests/test_distributed.py -k "whole_steps_in_one_launch or fused_stage_pair or peer_to_peer_halos or exchange_every_m_steps" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/tests.log | tail -20 | cut -c1-250
timeout 1200 python -m pytest -q -m gpu tests/test_gpu_parity.py -k "fused or triple" > $O/tests_parity.log 2>&1; echo "parity tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/tests_parity.log | tail -12 | cut -c1-250
rb() { timeout 400 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/${TAG}_rank.txt; }
for f in 2 auto 2 auto; do
  if [ $f = auto ]; then unset THETIS_AMD_FUSE12; else export THETIS_AMD_FUSE12=$f; fi
  echo "--- THETIS_AMD_FUSE12=$f" >> $O/${TAG}_rank.txt
  rb --case cfg2 --world 4 --rank 1 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
  rb --case cfg2 --world 2 --rank 0 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
  rb --case cfg2 --world 4 --rank 1 --every 2 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
  rb --case cfg2 --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
done
unset THETIS_AMD_FUSE12
sed 's/"exchange.*"rank"/ "rank"/; s/"every":/ every/; s/"overlap.*"fused_pair"/ "fused_pair"/' $O/${TAG}_rank.txt | cut -c1-200
timeout 600 python bench.py --no-cpu > $O/${TAG}_bench_line.json 2> $O/bench.err; tail -1 $O/${TAG}_bench_line.json | cut -c1-600
