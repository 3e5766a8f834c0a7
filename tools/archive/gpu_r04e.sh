#!/bin/bash
# round 4, fifth session: the exchange kernels on a side stream - bits, then ranks of 2 / 4 / 8 with and without
set -u
O=gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_distributed.py tests/test_gpu_spmd.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
rb() { timeout 300 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/rank.txt; }
for side in 0 1; do
  export THETIS_AMD_P2P_SIDE_STREAM=$side
  echo "side stream $side" >> $O/rank.txt
  rb --world 2 --rank 0 --every 4 --exchange p2p --flow 0 --graph-mode full --steps 960
  rb --world 2 --rank 0 --every 4 --exchange p2p --flow 0 --graph-mode cycle --steps 960
  rb --world 2 --rank 0 --every 8 --overlap 3 --exchange p2p --flow 0 --graph-mode full --steps 960
  rb --world 2 --rank 0 --every 2 --exchange p2p --flow 0 --graph-mode full --steps 960
  rb --world 4 --rank 1 --every 4 --exchange p2p --flow 0 --graph-mode full --steps 960
  rb --world 4 --rank 1 --every 4 --exchange p2p --flow 0 --graph-mode none --steps 960
  rb --world 4 --rank 1 --every 8 --overlap 3 --exchange p2p --flow 0 --graph-mode full --steps 960
  rb --world 4 --rank 1 --every 2 --exchange p2p --flow 0 --graph-mode full --steps 960
  rb --world 4 --rank 1 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
  rb --world 8 --rank 3 --every 4 --exchange p2p --flow 0 --graph-mode full --steps 1920
  rb --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
done
cut -c1-20,230- $O/rank.txt
THETIS_AMD_FLOW=0 timeout 300 python tools/kbench.py --steps 384 --prewarm 0.5 --tag single 2>&1 | tail -1 | cut -c1-200
