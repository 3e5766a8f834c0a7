#!/bin/bash
O=gpurun_out/r03i; mkdir -p $O
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
