#!/bin/bash
# closing evidence of round 4 with the library as committed: the whole GPU suite, the bench line, kernel-trace stats of the bench
# command, PMC traffic of the stage kernel at 1 M cells, every row of tools/cfgbench.py, ranks of 8 / 4 / 2 on one GPU
set -u
O=gpurun_out/evidence_r04; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-400
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -- python $R/bench.py --no-cpu > $R/$O/kstats.log 2>&1
cd $R
cp $(ls $O/kstats/*/*kernel_stats.csv | head -1) $O/r04z_kernel_stats.csv 2>/dev/null
head -5 $O/r04z_kernel_stats.csv | cut -c1-200
bash tools/pmc.sh $R/$O/pmc1m python $R/tools/kbench.py --steps 4 --order auto --calibrate > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc1m swe_ > $O/r04z_pmc_summary.txt 2>&1
python tools/make_traffic_json.py $O/pmc1m 1000000 $O/r04z_traffic.json "bench workload (1M triangles), round-4 library (stage kernel unchanged since r03a)" > /dev/null 2>&1
grep -E "traffic_bytes|algorithmic_bytes_per" $O/r04z_traffic.json
rm -rf $O/pmc1m
timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" > $O/r04z_cfgs.txt; cut -c1-200 $O/r04z_cfgs.txt
rb() { timeout 300 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/r04z_rank.txt; }
rb --world 8 --rank 3 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 1920
rb --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --world 8 --rank 0 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --world 4 --rank 1 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
rb --world 2 --rank 0 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
cut -c1-20,230- $O/r04z_rank.txt
find $O -name "*.csv" -size +3M -delete
find $O -name "*kernel_trace.csv" -delete
du -sh $O
