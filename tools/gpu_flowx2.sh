#!/bin/bash
O=gpurun_out/r03l; mkdir -p $O
for i in 1 2; do
timeout 1500 python -m pytest "tests/test_distributed.py::test_ranks_on_one_gpu_with_one_launch_per_cycle" -q -m gpu 2>&1 | tail -3
done
timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240 2>&1 | tail -1 >> $O/ab.log
timeout 300 python tools/rankbench.py --world 8 --rank 0 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240 2>&1 | tail -1 >> $O/ab.log
timeout 300 python tools/rankbench.py --world 16 --rank 7 --every 4 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240 2>&1 | tail -1 >> $O/ab.log
timeout 300 python tools/rankbench.py --world 16 --rank 7 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240 2>&1 | tail -1 >> $O/ab.log
timeout 300 python tools/rankbench.py --world 16 --rank 7 --every 4 --exchange p2p --nosplit --flow 0 --steps 240 2>&1 | tail -1 >> $O/ab.log
cat $O/ab.log
