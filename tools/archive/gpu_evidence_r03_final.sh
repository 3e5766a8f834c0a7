#!/bin/bash
# closing evidence of round 3 with the library as committed at its end: bench line, kernel-trace stats of the bench command, every
# row of tools/cfgbench.py (incl. general quadrilaterals), one rank of eight on the flow kernel
set -u
O=gpurun_out/evidence_r03m; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-300
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -- python $R/bench.py --no-cpu > $R/$O/kstats.log 2>&1
cd $R
cp $(ls $O/kstats/*/*kernel_stats.csv | head -1) $O/r03m_kernel_stats.csv 2>/dev/null
head -5 $O/r03m_kernel_stats.csv | cut -c1-200
timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" > $O/r03m_cfgs.txt; cut -c1-220 $O/r03m_cfgs.txt
rb() { timeout 300 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/r03m_rank.txt; }
rb --world 8 --rank 3 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 240
rb --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240
rb --world 8 --rank 0 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240
cut -c1-330 $O/r03m_rank.txt
find $O -name "*.csv" -size +3M -delete
find $O -name "*kernel_trace.csv" -delete
du -sh $O
