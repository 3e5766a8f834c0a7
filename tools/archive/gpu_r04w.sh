#!/bin/bash
# Round 4, session w: from which size does the LDS trace exchange (LDSX) pay, now that large launches alternate their direction?
mkdir -p gpurun_out/r04w
{
for size in "1118 559" "1225 612" "1414 707" "1581 790" "1732 866"; do set -- $size
  for rep in 1 2; do
  for x in 0 1; do
    THETIS_AMD_LDSX=$x python tools/kbench.py --nx $1 --ny $2 --steps 60 --prewarm 0.4 --tag "LDSX=$x"
  done; done
done
} 2>&1 | grep "^{" | cut -c1-150 > gpurun_out/r04w/ldsx_threshold.txt
cat gpurun_out/r04w/ldsx_threshold.txt
