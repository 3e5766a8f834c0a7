#!/bin/bash
# round 4, third GPU session: range check + the adversary of the granule protocol, the example scripts under several ranks,
# SIMD mates of the flow kernel, ranks of 2 / 4 / 8
set -u
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_spmd.py -q -m gpu -k "example" > $O/examples.log 2>&1; echo "examples rc=$?"; tail -3 $O/examples.log
bash tools/range_check.sh > $O/range_check.txt 2>&1; grep -E "rc=|passed|failed|violations|range check" $O/range_check.txt | tail -12
for nx in 125 354; do
  THETIS_AMD_LIB=$PWD/build_dbg/flow_wt.so timeout 300 python tools/flowtiming.py --nx $nx --ny $((nx/2)) > $O/flow_timing_$nx.json 2> $O/t.err
done
python - <<'PY'
import json
for nx in (125, 354):
    try:
        d = json.load(open('gpurun_out/r04c/flow_timing_%d.json' % nx))
        r = d['runs'][-1]
        print(nx, d['n_cells'], 'stage_us', round(r['stage_us'], 2), 'arith', round(r['arith_us'], 2), 'wait', round(r['wait_us'], 2), 'mates', json.dumps(r.get('mates')))
    except Exception as e:
        print(nx, 'failed', e)
PY
rb() { timeout 300 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/rank.txt; }
rb --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --world 2 --rank 0 --every 4 --exchange p2p --flow 0 --graph-mode full --steps 960
rb --world 2 --rank 0 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
rb --world 2 --rank 0 --every 8 --overlap 3 --exchange p2p --flow 0 --graph-mode cycle --steps 960
rb --world 2 --rank 0 --every 2 --exchange p2p --flow 0 --graph-mode full --steps 960
rb --world 2 --rank 0 --every 1 --exchange p2p --flow 0 --graph-mode full --steps 960
rb --world 4 --rank 1 --every 4 --exchange p2p --flow 0 --graph-mode full --steps 960
rb --world 4 --rank 1 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
rb --world 4 --rank 1 --every 8 --overlap 3 --exchange p2p --flow 0 --graph-mode cycle --steps 960
rb --world 4 --rank 1 --every 2 --exchange p2p --flow 0 --graph-mode full --steps 960
cut -c1-20,230- $O/rank.txt
for nx in 707 500; do THETIS_AMD_FLOW=0 timeout 300 python tools/kbench.py --nx $nx --ny 500 --steps 384 --prewarm 0.5 --tag single 2>&1 | tail -1 >> $O/sizes.txt; done
THETIS_AMD_FLOW=0 timeout 300 python tools/kbench.py --nx 250 --ny 500 --steps 384 --prewarm 0.5 --tag single 2>&1 | tail -1 >> $O/sizes.txt
cat $O/sizes.txt | cut -c1-300
