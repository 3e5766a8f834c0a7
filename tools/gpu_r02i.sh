#!/bin/bash
set -u
O=gpurun_out/r02i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for nx in 250 500 1000 2000 4000; do
for b in 0 1; do
  THETIS_AMD_PERSISTENT=$b timeout 300 python tools/kbench.py --nx $nx --ny 500 --tag "pers$b" 2>/dev/null | tail -1 >> $O/kbench.log
done
done
THETIS_AMD_PERSISTENT=1 THETIS_AMD_PERSISTENT_WAVES=1024 timeout 300 python tools/kbench.py --nx 1000 --ny 500 --tag "pers1_w1024" 2>/dev/null | tail -1 >> $O/kbench.log
THETIS_AMD_PERSISTENT=1 THETIS_AMD_PERSISTENT_WAVES=3072 timeout 300 python tools/kbench.py --nx 1000 --ny 500 --tag "pers1_w3072" 2>/dev/null | tail -1 >> $O/kbench.log
cat $O/kbench.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['n_cells'], d['tag'], round(d['us_per_step'],2), round(d['us_per_launch'],2), round(d['frac'],3))
"
