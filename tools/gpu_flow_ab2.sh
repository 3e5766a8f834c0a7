#!/bin/bash
O=gpurun_out/r03d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_flow_kernel.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -15 $O/tests.log
for nx in 125 250 354 360; do
  THETIS_AMD_FLOW=1 THETIS_AMD_FUSED_STEP=0 timeout 300 python tools/kbench.py --nx $nx --ny $((nx/2)) --steps 96 --tag flow2 2>&1 | tail -1 >> $O/ab.log
done
THETIS_AMD_FLOW=1 THETIS_AMD_FUSED_STEP=0 timeout 300 python tools/kbench.py --nx 256 --ny 256 --steps 96 --tag flow2_2048blocks 2>&1 | tail -1 >> $O/ab.log
timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --steps 240 2>&1 | tail -1 >> $O/ab.log
timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --graph-mode full --steps 240 2>&1 | tail -1 >> $O/ab.log
cat $O/ab.log
for nx in 125 354; do
    echo "== flow_wt nx=$nx" >> $O/timing.log
    THETIS_AMD_LIB=$PWD/variants/flow_wt.so timeout 300 python tools/flowtiming.py --nx $nx --ny $((nx/2)) > $O/t.json 2> $O/t.err
    python - >> $O/timing.log <<PY
import json
try:
    d = json.load(open('$O/t.json'))
    for r in d['runs'][1:4]: print(json.dumps(r))
except Exception as e:
    print('failed', e); print(open('$O/t.err').read()[-2000:]); print(open('$O/t.json').read()[:500])
PY
done
cat $O/timing.log
