#!/bin/bash
# round 2 closing evidence with the final library: bench, kernel-trace stats of the bench command, PMC traffic (1M, 4M)
set -u
O=gpurun_out/evidence_bench; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -- python $R/bench.py --no-cpu --no-beyond-cache > $R/$O/kstats.log 2>&1
cd $R
cp $(ls $O/kstats/*/*kernel_stats.csv | head -1) $O/r02f_kernel_stats.csv 2>/dev/null
head -5 $O/r02f_kernel_stats.csv; tail -1 $O/kstats.log
bash tools/pmc.sh $R/$O/pmc1m python $R/tools/kbench.py --steps 4 --order auto --calibrate > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc1m swe_ > $O/r02f_pmc_summary.txt 2>&1
python tools/make_traffic_json.py $O/pmc1m 1000000 $O/r02f_traffic.json "bench workload (1M triangles), final round-2 stage kernel (boundary-inline variant, traces gathered from memory; 164 VGPRs, no scratch)" > /dev/null 2>&1
bash tools/pmc.sh $R/$O/pmc4m python $R/tools/kbench.py --steps 3 --order auto --calibrate --nx 2000 --ny 1000 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc4m swe_ > $O/r02f_pmc_summary_4m.txt 2>&1
python tools/make_traffic_json.py $O/pmc4m 4000000 $O/r02f_traffic_4m.json "4M triangles (beyond the Infinity Cache), final round-2 stage kernel (boundary-inline + LDS trace exchange variant)" > /dev/null 2>&1
grep -E "traffic_bytes|algorithmic_bytes_per" $O/r02f_traffic.json $O/r02f_traffic_4m.json
find $O -name "*.csv" -size +3M -delete
du -sh $O
