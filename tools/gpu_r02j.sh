#!/bin/bash
set -u
O=gpurun_out/r02j; mkdir -p $O
for nx in 1000 2000 4000; do
for b in 0 1; do
  THETIS_AMD_BND_INLINE=$b timeout 300 python tools/kbench.py --nx $nx --ny 500 --tag "binl$b" 2>/dev/null | tail -1 >> $O/kbench.log
done
done
THETIS_AMD_LIB=$PWD/variants/minw3.so THETIS_AMD_BND_INLINE=1 timeout 300 python tools/kbench.py --nx 4000 --ny 500 --tag "minw3_binl1" 2>/dev/null | tail -1 >> $O/kbench.log
cat $O/kbench.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['n_cells'], d['tag'], round(d['us_per_step'],2), round(d['us_per_launch'],2), round(d['frac'],3))
"
