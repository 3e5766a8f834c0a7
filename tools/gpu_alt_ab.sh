#!/bin/bash
O=gpurun_out/r03h; mkdir -p $O
for sz in "1000 500" "1414 707" "2000 1000" "2828 1414"; do
  set -- $sz
  for alt in 0 1; do
    THETIS_AMD_ALTERNATE=$alt timeout 300 python tools/kbench.py --nx $1 --ny $2 --steps 30 --prewarm 0.5 --tag alt$alt 2>&1 | tail -1 >> $O/alt.log
  done
done
cat $O/alt.log
