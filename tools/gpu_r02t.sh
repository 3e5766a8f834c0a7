#!/bin/bash
set -u
O=gpurun_out/r02t; mkdir -p $O
for rep in 1 2; do
for sz in "250 500" "500 500" "1000 500" "2000 500" "2000 1000"; do
  set -- $sz
  for l in 0 1; do
    THETIS_AMD_LDSX=$l timeout 300 python tools/kbench.py --order auto --nx $1 --ny $2 --prewarm 0.5 --tag "ldsx$l" 2>/dev/null | tail -1 >> $O/kbench.log
  done
done
done
THETIS_AMD_BND_INLINE=0 THETIS_AMD_LDSX=0 timeout 300 python tools/kbench.py --order auto --nx 1000 --ny 500 --prewarm 0.5 --tag "epilogue_noldsx" 2>/dev/null | tail -1 >> $O/kbench.log
THETIS_AMD_BND_INLINE=0 THETIS_AMD_LDSX=0 timeout 300 python tools/kbench.py --order auto --nx 2000 --ny 1000 --prewarm 0.5 --tag "epilogue_noldsx" 2>/dev/null | tail -1 >> $O/kbench.log
cat $O/kbench.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['n_cells'], d['tag'], round(d['us_per_step'],2), round(d['frac'],3))
"
