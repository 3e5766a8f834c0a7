#!/bin/bash
# one rank of eight / sixteen of the bench mesh on this GPU (loopback peers): step times of the schedules, and the per-block time
# stamps of a normal stage (S = 8) and of the first stage of a cycle (S = 6) by the block's role (tools/rankbench.py --timing with
# -DSWE_WAVE_TIMING -DSWE_FLOW_TS_STAGE=S builds of the library in variants/)
O=gpurun_out/r03t; mkdir -p $O; rm -f $O/*.txt
rb() { timeout 300 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/r03o_rank.txt; }
rb --world 8 --rank 3 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 1920
rb --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 0 --graph-mode full --steps 1920
rb --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --steps 1920
rb --world 8 --rank 0 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --world 8 --rank 3 --every 1 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --world 16 --rank 7 --every 4 --exchange p2p --nosplit --flow 0 --steps 1920
rb --world 16 --rank 7 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --world 16 --rank 7 --every 4 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
cut -c1-20,250- $O/r03o_rank.txt
for S in 6 8; do
  THETIS_AMD_LIB=$PWD/variants/wt$S.so timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 --timing 2>&1 | tail -2 | sed "s/^/stage $S: /" >> $O/r03o_rank_timing.txt
done
# the flow path on one device at five sizes, against the stage launches
for nx in 125 250 354 358 360 362; do for fl in 0 1; do
  THETIS_AMD_FLOW=$fl timeout 300 python tools/kbench.py --nx $nx --ny $((nx/2)) --steps 384 --prewarm 0.5 --tag flow$fl 2>&1 | tail -1 >> $O/r03o_flow_sizes.txt
done; done
for nx in 125 354; do
  THETIS_AMD_LIB=$PWD/variants/flow_wt.so timeout 300 python tools/flowtiming.py --nx $nx --ny $((nx/2)) > $O/r03o_flow_timing_$nx.json 2> $O/t.err
done
