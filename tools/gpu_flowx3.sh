#!/bin/bash
O=gpurun_out/r03n; mkdir -p $O
for nc in 1 2 4 8; do
THETIS_AMD_FLOWX_CYCLES=$nc timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240 2>&1 | tail -1 | sed "s/^/cycles_per_launch=$nc /" >> $O/ab.log
done
THETIS_AMD_FLOWX_CYCLES=16 timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 1 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240 2>&1 | tail -1 | sed "s/^/m1 /" >> $O/ab.log
cat $O/ab.log
