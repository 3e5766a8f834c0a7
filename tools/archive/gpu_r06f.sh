#!/bin/bash
# round 6, sixth run: the fused kernels with self-contained tile records (one dependent trip to memory less per tile): bitwise tests,
# sizes 250 k ... 4 M, the bench line; cfg 4 ranks with the limiter's means inside the vertex kernel
set -u
TAG=r06f
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest -q -m gpu tests/test_gpu_parity.py -k "fused or stage_solutions" tests/test_distributed.py -k "fused_stage_pair or tracer or coupled" tests/test_gpu_tracer.py > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -10 | cut -c1-250
kb() { timeout 300 python tools/kbench.py "$@" 2>&1 | tail -1 >> $O/${TAG}_fused_sizes.txt; }
for sz in "500 250" "707 354" "1000 500" "2000 1000"; do
  set -- $sz
  kb --nx $1 --ny $2 --steps 40 --prewarm 0.5 --tag auto
done
THETIS_AMD_FUSE12=3 kb --nx 1000 --ny 500 --steps 40 --prewarm 0.5 --tag fuse3
THETIS_AMD_FUSE12=0 kb --nx 1000 --ny 500 --steps 40 --prewarm 0.5 --tag fuse0
python - <<'PY'
import json
for l in open('gpurun_out/r06f/r06f_fused_sizes.txt'):
    d = json.loads(l); print(d['tag'], d['n_cells'], '%.2f us/step' % d['us_per_step'], 'frac %.3f' % d['frac'])
PY
timeout 600 python bench.py --no-cpu > $O/${TAG}_bench_line.json 2> $O/bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06f/r06f_bench_line.json').read().strip().splitlines()[-1])
r = d['roofline']
print('ms/step', d['ms_per_step'], 'frac', r['frac'], r['frac_samples'], 'beyond', r['frac_beyond_cache'])
PY
rb() { timeout 400 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/${TAG}_rank.txt; }
rb --case cfg4 --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480
rb --case cfg4_tracer_only --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480
rb --case cfg4 --world 8 --rank 3 --every 4 --exchange p2p --graph-mode full --steps 480
rb --case cfg2 --world 4 --rank 1 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
rb --case cfg2 --world 2 --rank 0 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
python - <<'PY'
import json
for l in open('gpurun_out/r06f/r06f_rank.txt'):
    try: d = json.loads(l)
    except Exception: print(l[:200]); continue
    print(d['case'], 'world', d['world'], 'every', d['every'], 'fused', d['fused_pair'][:1], 'us/step %.2f' % d['us_per_step'])
PY
du -sh $O
