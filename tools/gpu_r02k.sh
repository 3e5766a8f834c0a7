#!/bin/bash
# round 2 validation + profiles: full GPU suite, bench (with cpu baseline + beyond-cache), kernel-trace stats, PMC traffic (1M, 4M)
set -u
O=gpurun_out/r02k; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?" >> $O/pytest_all.log
tail -4 $O/pytest_all.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -- python $R/bench.py --no-cpu --no-beyond-cache > $R/$O/kstats.log 2>&1
cd $R
cp $(ls $O/kstats/*/*kernel_stats.csv | head -1) $O/r02a_kernel_stats.csv 2>/dev/null
head -6 $O/r02a_kernel_stats.csv
tail -1 $O/kstats.log
bash tools/pmc.sh $R/$O/pmc1m python $R/tools/kbench.py --steps 4 --order auto --calibrate > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc1m swe_ > $O/r02a_pmc_summary.txt 2>&1
python tools/make_traffic_json.py $O/pmc1m 1000000 $O/r02a_traffic.json "bench workload (1M triangles), round-2 stage kernel (boundary-inline variant)" > /dev/null 2>&1
bash tools/pmc.sh $R/$O/pmc4m python $R/tools/kbench.py --steps 3 --order auto --calibrate --nx 2000 --ny 1000 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc4m swe_ > $O/r02b_pmc_summary_4m.txt 2>&1
python tools/make_traffic_json.py $O/pmc4m 4000000 $O/r02b_traffic_4m.json "4M triangles (beyond the Infinity Cache), round-2 stage kernel" > /dev/null 2>&1
grep -E "traffic_bytes|algorithmic_bytes_per" $O/r02a_traffic.json $O/r02b_traffic_4m.json
find $O -name "*.csv" -size +3M -delete
du -sh $O
