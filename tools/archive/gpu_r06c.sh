#!/bin/bash
# round 6, third run: what a rank of eight of cfg 4 spends its step on (kernel stats), the fused pair forced on ranks of eight
# (partitions below the 250 k rule, where the dataflow kernel does not apply: cfg 4, m = 4), the bench line with the size rule for
# the three-stage kernel at 4 M cells and its PMC traffic there
set -u
TAG=r06c
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest -q -m gpu tests/test_distributed.py -k "fused_stage_pair" tests/test_gpu_parity.py -k "fused" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -5 | cut -c1-250
rb() { timeout 400 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/${TAG}_rank.txt; }
for f in 0 1; do
  export THETIS_AMD_FUSE12=$f
  rb --case cfg4 --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480
  rb --case cfg4_tracer_only --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480
  rb --case cfg2 --world 8 --rank 3 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
done
unset THETIS_AMD_FUSE12
python - <<'PY'
import json
for l in open('gpurun_out/r06c/r06c_rank.txt'):
    try: d = json.loads(l)
    except Exception: print(l[:200]); continue
    print(d['case'], 'world', d['world'], 'every', d['every'], 'fused', d['fused_pair'][:3], 'us/step %.2f' % d['us_per_step'])
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats_cfg4 -- python $R/tools/rankbench.py --case cfg4 --world 8 --rank 3 --every 2 --exchange p2p --graph-mode none --steps 240 > $R/$O/kstats_cfg4.log 2>&1
cd $R
cp $(ls $O/kstats_cfg4/*/*kernel_stats.csv | head -1) $O/${TAG}_cfg4_rank8_kernel_stats.csv 2>/dev/null
cut -d, -f1-4 $O/${TAG}_cfg4_rank8_kernel_stats.csv | cut -c1-150 | head -14
find $O -name "*kernel_trace.csv" -delete
timeout 900 python bench.py --no-cpu > $O/${TAG}_bench_line.json 2> $O/bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06c/r06c_bench_line.json').read().strip().splitlines()[-1])
r = d['roofline']
print('ms/step', d['ms_per_step'], 'frac', r['frac'], 'fused_model', r['frac_fused_model'], 'traffic_frac', r['traffic_rate_frac'], 'valu', r['valu_issue_frac'])
print('beyond', r['frac_beyond_cache'], r['beyond_cache'])
PY
bash tools/pmc.sh $R/$O/pmc4m python $R/tools/kbench.py --nx 2000 --ny 1000 --steps 3 --prewarm 0.2 --order auto --calibrate > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc4m swe_ > $O/${TAG}_pmc_summary_4m.txt 2>&1
python tools/make_traffic_json.py $O/pmc4m 4000000 $O/${TAG}_traffic_4m.json "RectangleMesh(2000,1000) = 4M triangles (roofline.beyond_cache), all three stages in one launch ($TAG)" > /dev/null 2>&1
grep -E "traffic_bytes|launches_per_step|valu_wave|\"name\"" $O/${TAG}_traffic_4m.json
rm -rf $O/pmc4m
du -sh $O
