#!/bin/bash
# closing evidence of round 5 with the library as committed: the whole GPU suite, the adversary builds (tools/range_check.sh covers
# them too), the bench line, kernel-trace stats of the bench command and of cfg 5, PMC traffic of the stage kernel at 1 M cells, PMC
# instruction counts of cfg 5, every row of tools/cfgbench.py, ranks of 8 / 4 / 2 on one GPU incl. cfg 4 / cfg 5 strips
# TAG (default r05z): prefix of the files it leaves; the closing set after the 16-B connectivity records: TAG=r05zz
set -u
TAG=${TAG:-r05z}
O=gpurun_out/evidence_$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-400
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -- python $R/bench.py --no-cpu > $R/$O/kstats.log 2>&1
CFGBENCH_ONLY=cfg5 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats5 -- python $R/tools/cfgbench.py > $R/$O/kstats5.log 2>&1
cd $R
cp $(ls $O/kstats/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv 2>/dev/null
cp $(ls $O/kstats5/*/*kernel_stats.csv | head -1) $O/${TAG}_cfg5_kernel_stats.csv 2>/dev/null
head -5 $O/${TAG}_kernel_stats.csv | cut -c1-200; head -4 $O/${TAG}_cfg5_kernel_stats.csv | cut -c1-200
bash tools/pmc.sh $R/$O/pmc1m python $R/tools/kbench.py --steps 4 --order auto --calibrate > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc1m swe_ > $O/${TAG}_pmc_summary.txt 2>&1
python tools/make_traffic_json.py $O/pmc1m 1000000 $O/${TAG}_traffic.json "bench workload (1M triangles), round-5 library ($TAG)" > /dev/null 2>&1
grep -E "traffic_bytes|algorithmic_bytes_per" $O/${TAG}_traffic.json
rm -rf $O/pmc1m
CFGBENCH_ONLY=cfg5_profile bash tools/pmc.sh $R/$O/pmc5 python $R/tools/cfgbench.py > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc5 swe_stage_kernel > $O/${TAG}_cfg5_pmc_summary.txt 2>&1
grep -E "^swe_stage|SQ_INSTS_VALU |SQ_WAVES " $O/${TAG}_cfg5_pmc_summary.txt | cut -c1-160
rm -rf $O/pmc5
timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" > $O/${TAG}_cfgs.txt; cut -c1-200 $O/${TAG}_cfgs.txt
rb() { timeout 400 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/${TAG}_rank.txt; }
rb --case cfg2 --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --case cfg2 --world 8 --rank 0 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --case cfg2 --world 8 --rank 3 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 1920
rb --case cfg2 --world 4 --rank 1 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
rb --case cfg2 --world 2 --rank 0 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
rb --case cfg4 --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480
rb --case cfg4_tracer_only --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480
rb --case cfg2_src --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --case cfg5 --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 960
rb --case cfg5 --world 8 --rank 7 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 960
rb --case cfg5 --world 8 --rank 3 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
cut -c1-20,230- $O/${TAG}_rank.txt
find $O -name "*.csv" -size +3M -delete
find $O -name "*kernel_trace.csv" -delete
du -sh $O
