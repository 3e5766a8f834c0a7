#!/bin/bash
# round 2 final evidence: full GPU suite, bench, kernel-trace stats, PMC traffic (1M, 4M), configuration table, rank bench, sweep
set -u
O=gpurun_out/evidence_cfgs; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?" >> $O/pytest_all.log
tail -3 $O/pytest_all.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -- python $R/bench.py --no-cpu --no-beyond-cache > $R/$O/kstats.log 2>&1
cd $R
cp $(ls $O/kstats/*/*kernel_stats.csv | head -1) $O/r02c_kernel_stats.csv 2>/dev/null
head -4 $O/r02c_kernel_stats.csv; grep -o '"ms_per_step": [0-9.]*' $O/kstats.log | tail -1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats_cfg -- python $R/tools/cfgbench.py > $R/$O/cfgbench_prof.log 2>&1
cd $R
cp $(ls $O/kstats_cfg/*/*kernel_stats.csv | head -1) $O/r02d_kernel_stats_cfgs.csv 2>/dev/null
timeout 600 python tools/cfgbench.py 2>/dev/null > $O/cfgbench.log; cat $O/cfgbench.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['config'][:70], d['n_cells'], round(d['us_per_step'],1), round(d['frac_of_8TBs'],3))
"
bash tools/pmc.sh $R/$O/pmc1m python $R/tools/kbench.py --steps 4 --order auto --calibrate > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc1m swe_ > $O/r02c_pmc_summary.txt 2>&1
python tools/make_traffic_json.py $O/pmc1m 1000000 $O/r02c_traffic.json "bench workload (1M triangles), round-2 final stage kernel (boundary-inline + LDS exchange variant, 164 VGPRs, no scratch)" > /dev/null 2>&1
bash tools/pmc.sh $R/$O/pmc4m python $R/tools/kbench.py --steps 3 --order auto --calibrate --nx 2000 --ny 1000 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc4m swe_ > $O/r02e_pmc_summary_4m.txt 2>&1
python tools/make_traffic_json.py $O/pmc4m 4000000 $O/r02e_traffic_4m.json "4M triangles (beyond the Infinity Cache), round-2 final stage kernel" > /dev/null 2>&1
grep -E "traffic_bytes|algorithmic_bytes_per" $O/r02c_traffic.json $O/r02e_traffic_4m.json
bash tools/pmc.sh $R/$O/pmccfg python $R/tools/cfgbench.py > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmccfg swe_ > $O/r02f_pmc_summary_cfgs.txt 2>&1
for a in "--every 4" "--every 4 --exchange p2p" "--every 4 --exchange p2p --nosplit" "--every 8 --exchange p2p --nosplit" "--every 2 --exchange p2p --nosplit"; do
  timeout 300 python tools/rankbench.py --world 8 --rank 3 $a 2>/dev/null | tail -1 >> $O/rankbench.log
done
for w in 2 4; do timeout 300 python tools/rankbench.py --world $w --rank 1 --every 4 --exchange p2p --nosplit 2>/dev/null | tail -1 >> $O/rankbench.log; done
cat $O/rankbench.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['world'], d['exchange'], 'split' if d['split'] else 'nosplit', 'm', d['every'], round(d['us_per_step'],2))
"
for sz in "62 500" "125 500" "250 500" "500 500" "1000 500" "2000 500" "2000 1000" "4000 1000"; do
  set -- $sz
  timeout 300 python tools/kbench.py --nx $1 --ny $2 --tag sweep --prewarm 0.5 2>/dev/null | tail -1 >> $O/kbench_sweep.log
done
cat $O/kbench_sweep.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['n_cells'], round(d['us_per_step'],2), round(d['frac'],3))
"
THETIS_AMD_LIB=$PWD/variants/wt.so timeout 300 python tools/wavetiming.py --nx 125 --ny 500 --stage 1 2>/dev/null > $O/wt_125k.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02r/wt_125k.json')); print(json.dumps(d['runs'][-1]))
PY
timeout 300 python tools/unstructured_bench.py 2>/dev/null | tail -1 > $O/unstructured.log; cut -c1-400 $O/unstructured.log
find $O -name "*.csv" -size +2M -delete
du -sh $O
