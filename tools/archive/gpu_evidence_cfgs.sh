#!/bin/bash
# closing evidence of the round with the final library: every configuration of the path at bench size (wall time + kernel-trace
# stats), one rank of 2 / 4 / 8 with the peer-to-peer halo (loopback), and the size sweep of the headline kernel
set -u
O=gpurun_out/evidence_cfgs; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
echo "# tools/cfgbench.py (plain run, no profiler): configuration, cells, us/step, fraction of 8 TB/s of the algorithmic bytes"
timeout 600 python tools/cfgbench.py 2>/dev/null | grep '^{'
echo "# tools/rankbench.py: one rank of N on one GPU, peers = loopback, graph-replayed cycles, us/step"
for a in "--world 8 --rank 3 --every 4" "--world 8 --rank 3 --every 4 --exchange p2p" "--world 8 --rank 3 --every 4 --exchange p2p --nosplit" \
         "--world 8 --rank 3 --every 8 --exchange p2p --nosplit" "--world 8 --rank 3 --every 2 --exchange p2p --nosplit" \
         "--world 4 --rank 1 --every 4 --exchange p2p --nosplit" "--world 2 --rank 1 --every 4 --exchange p2p --nosplit"; do
  timeout 300 python tools/rankbench.py $a 2>/dev/null | tail -1
done
echo "# tools/kbench.py size sweep (production numbering; <= 80 k cells: swe2d_advance takes the step kernel)"
for sz in "250 125" "354 177" "500 250" "707 354" "1000 500" "1414 707" "2000 1000" "2828 1414"; do
  set -- $sz
  timeout 300 python tools/kbench.py --nx $1 --ny $2 --tag sweep --prewarm 0.5 2>/dev/null | tail -1
done
} > $O/r02h_cfgs_ranks_sweep.txt
cut -c1-200 $O/r02h_cfgs_ranks_sweep.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats_cfg -- python $R/tools/cfgbench.py > $R/$O/cfgbench_prof.log 2>&1
cd $R
cp $(ls $O/kstats_cfg/*/*kernel_stats.csv | head -1) $O/r02h_kernel_stats_cfgs.csv 2>/dev/null; head -12 $O/r02h_kernel_stats_cfgs.csv | cut -c1-160
find $O -name "*.csv" -size +3M -delete
