#!/bin/bash
O=gpurun_out/r03k; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_flow_kernel.py "tests/test_distributed.py::test_ranks_on_one_gpu_with_one_launch_per_cycle" -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -25 $O/tests.log | cut -c1-300
for nx in 125 354; do
  THETIS_AMD_FLOW=1 timeout 300 python tools/kbench.py --nx $nx --ny $((nx/2)) --steps 96 --tag flow 2>&1 | tail -1 >> $O/ab.log
done
for m in 1 2; do
for fx in 0 1; do
timeout 300 python tools/rankbench.py --world 8 --rank 3 --every $m --exchange p2p --nosplit --flow 1 --flowx $fx --steps 240 2>&1 | tail -1 >> $O/ab.log
done
done
timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240 2>&1 | tail -1 >> $O/ab.log
timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode none --steps 240 2>&1 | tail -1 >> $O/ab.log
cat $O/ab.log
