#!/bin/bash
# round 5, first contact: the whole GPU suite with the split library, the bench line, baselines of the ranks (incl. the per-block
# time stamps of the flow kernel: build_dbg/wt6.so, wt8.so = -DSWE_WAVE_TIMING -DSWE_FLOW_TS_STAGE=S unity builds), cfg rows
set -u
O=gpurun_out/r05a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-300
rb() { timeout 300 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/rank.txt; }
rb --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --world 8 --rank 0 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --world 8 --rank 3 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 1920
rb --world 4 --rank 1 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
rb --world 2 --rank 0 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
cut -c1-20,230- $O/rank.txt
for S in 6 8; do
  THETIS_AMD_LIB=$PWD/build_dbg/wt$S.so RANKBENCH_TIMING_DUMP=$O/stamps_$S.npz timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 --timing 2>&1 | tail -2 | sed "s/^/stage $S: /" >> $O/rank_timing.txt
done
cat $O/rank_timing.txt | cut -c1-1500
for nx in 125 354; do
  THETIS_AMD_LIB=$PWD/build_dbg/wt8.so timeout 300 python tools/flowtiming.py --nx $nx --ny $((nx/2)) > $O/flow_timing_$nx.json 2> $O/t.err
done
timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" > $O/cfgs.txt; cut -c1-200 $O/cfgs.txt
du -sh $O
