#!/bin/bash
# round 5, run q: wetting-drying in the flow kernel (swe_flow_kernel<..., WD>): bitwise tests, wetting-drying / distributed / spmd /
# example tests, cfg 5 rank rows with the flow kernel (exchange kernels after the launch / inside it) against stage launches, a small
# Balzano mesh on one device
set -u
O=gpurun_out/r05q; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests/test_gpu_flow_kernel.py tests/test_wetting_drying.py tests/test_distributed.py tests/test_gpu_spmd.py tests/test_gpu_examples.py tests/test_gpu_fuzz.py tests/test_gpu_solver2d.py -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -6 $O/gpu_tests.log | cut -c1-400
rb() { timeout 400 python tools/rankbench.py --case cfg5 --world 8 --rank $1 --every $2 --exchange p2p --nosplit --flow $3 --flowx $4 --graph-mode full --steps 960 2>&1 | tail -1; }
for rep in 1 2; do
  rb 3 2 0 0 >> $O/rank.txt
  rb 3 2 1 0 >> $O/rank.txt
  rb 3 2 1 1 >> $O/rank.txt
  rb 3 4 0 0 >> $O/rank.txt
  rb 3 4 1 0 >> $O/rank.txt
  rb 7 2 1 1 >> $O/rank.txt
done
sed 's/"exchange.*"world"/ world/; s/"overlap.*"n_owned"/ n_owned/; s/"n_send.*"flow"/ flow/' $O/rank.txt
