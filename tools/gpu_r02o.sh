#!/bin/bash
set -u
O=gpurun_out/r02o; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py tests/test_gpu_sipg.py tests/test_unstructured.py tests/test_wetting_drying.py tests/test_gpu_solver2d.py -m gpu -x -q 2>&1 | tail -3
for sz in "125 500" "250 500" "500 500" "1000 500" "2000 1000"; do
  set -- $sz
  timeout 300 python tools/kbench.py --nx $1 --ny $2 --tag "auto" 2>/dev/null | tail -1 >> $O/kbench.log
  THETIS_AMD_LDSX=0 timeout 300 python tools/kbench.py --nx $1 --ny $2 --tag "ldsx0" 2>/dev/null | tail -1 >> $O/kbench.log
  THETIS_AMD_LDSX=1 timeout 300 python tools/kbench.py --nx $1 --ny $2 --tag "ldsx1" 2>/dev/null | tail -1 >> $O/kbench.log
done
cat $O/kbench.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['n_cells'], d['tag'], round(d['us_per_step'],2), round(d['us_per_launch'],2), round(d['frac'],3), d['vol'])
"
timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 4 --exchange p2p --nosplit 2>/dev/null | tail -1
