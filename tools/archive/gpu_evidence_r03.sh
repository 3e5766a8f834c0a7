#!/bin/bash
# round 3 closing evidence with the library as committed: bench line, kernel-trace stats of the bench command, PMC traffic at
# 1 M / 2 M / 4 M triangles, the flow kernel (sizes, per-block time stamps, kernel trace), one rank of eight / sixteen on the stage
# launches and on the flow kernel without / with the exchange inside, the other configurations
set -u
O=gpurun_out/evidence_r03; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-400
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -- python $R/bench.py --no-cpu > $R/$O/kstats.log 2>&1
cd $R
cp $(ls $O/kstats/*/*kernel_stats.csv | head -1) $O/r03a_kernel_stats.csv 2>/dev/null
head -5 $O/r03a_kernel_stats.csv | cut -c1-200
# PMC traffic per size (separate --pmc passes, kernel trace only)
for sz in "1000 500 1000000 r03a_traffic.json" "1414 707 1999396 r03a_traffic_2m.json" "2000 1000 4000000 r03a_traffic_4m.json"; do
  set -- $sz
  bash tools/pmc.sh $R/$O/pmc_$3 python $R/tools/kbench.py --steps 4 --order auto --calibrate --nx $1 --ny $2 > /dev/null 2>&1
  cd $R
  python tools/pmc_summary.py $O/pmc_$3 swe_ > $O/${4%.json}_pmc_summary.txt 2>&1
  python tools/make_traffic_json.py $O/pmc_$3 $3 $O/$4 "round-3 stage kernel ($3 triangles; launches of at least 256 MB of state alternate their direction)" > /dev/null 2>&1
done
grep -E "traffic_bytes|algorithmic_bytes_per" $O/r03a_traffic*.json
# flow kernel: sizes, stamps, trace
for nx in 125 250 354 362; do
  for fl in 0 1; do
    THETIS_AMD_FLOW=$fl timeout 300 python tools/kbench.py --nx $nx --ny $((nx/2)) --steps 96 --prewarm 0.5 --tag flow$fl 2>&1 | tail -1 >> $O/r03f_flow_sizes.txt
  done
done
cat $O/r03f_flow_sizes.txt | cut -c1-200
for nx in 125 354; do
  THETIS_AMD_LIB=$R/variants/flow_wt.so timeout 300 python tools/flowtiming.py --nx $nx --ny $((nx/2)) > $O/r03b_flow_timing_$nx.json 2> $O/t.err
done
cd /tmp
THETIS_AMD_FLOW=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kflow -- python $R/tools/kbench.py --nx 354 --ny 177 --steps 96 > $R/$O/kflow.log 2>&1
cd $R
cp $(ls $O/kflow/*/*kernel_stats.csv | head -1) $O/r03g_flow_kernel_stats.csv 2>/dev/null
head -4 $O/r03g_flow_kernel_stats.csv | cut -c1-200
# one rank of a strong-scaling run on this GPU (loopback peers)
rb() { timeout 300 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/r03e_rank_flow.txt; }
rb --world 8 --rank 3 --every 4 --exchange p2p --nosplit --flow 0 --steps 240
rb --world 8 --rank 3 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 240
rb --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 0 --graph-mode full --steps 240
rb --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240
rb --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --steps 240
rb --world 8 --rank 0 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240
rb --world 8 --rank 3 --every 1 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240
rb --world 16 --rank 7 --every 4 --exchange p2p --nosplit --flow 0 --steps 240
rb --world 16 --rank 7 --every 4 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240
cat $O/r03e_rank_flow.txt | cut -c1-330
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/krank -- python $R/tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode none --steps 240 > $R/$O/krank.log 2>&1
cd $R
cp $(ls $O/krank/*/*kernel_stats.csv | head -1) $O/r03c_rank_fx_kernel_stats.csv 2>/dev/null
head -4 $O/r03c_rank_fx_kernel_stats.csv | cut -c1-200
timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" > $O/r03h_cfgs.txt; cut -c1-200 $O/r03h_cfgs.txt
find $O -name "*.csv" -size +3M -delete
find $O -name "*kernel_trace.csv" -delete
du -sh $O
