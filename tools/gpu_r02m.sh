#!/bin/bash
set -u
O=gpurun_out/r02m; mkdir -p $O
for nx in 62 125 250 1000 4000; do
  timeout 300 python tools/kbench.py --nx $nx --ny 500 --tag "base" 2>/dev/null | tail -1 >> $O/kbench.log
  THETIS_AMD_LIB=$PWD/variants/ldsx.so timeout 300 python tools/kbench.py --nx $nx --ny 500 --tag "ldsx" 2>/dev/null | tail -1 >> $O/kbench.log
done
cat $O/kbench.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['n_cells'], d['tag'], round(d['us_per_step'],2), round(d['us_per_launch'],2), round(d['frac'],3), d['vol'])
"
