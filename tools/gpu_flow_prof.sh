#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode none --steps 240 > $O/rank_prof.log 2>&1
grep us_per_step $O/rank_prof.log | cut -c1-300
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/rank_fx_kernel_stats.csv
head -6 $O/rank_fx_kernel_stats.csv | cut -c1-220
python $GRAFT_REPO_ROOT/tools/timeline.py $(ls $O/prof/*/*kernel_trace.csv | head -1) --last 60 2>&1 | tail -22
find $O -name "*.csv" -size +3M -delete
