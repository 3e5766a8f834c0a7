#!/bin/bash
set -u
O=gpurun_out/r04k; mkdir -p $O
export TMPDIR=/tmp
for a in "--nx 354 --ny 177 --plain --steps 1500" "--nx 707 --ny 354 --steps 1500"; do
  timeout 300 python tools/forcingbench.py $a 2>&1 | tail -1 | tee -a $O/forcing2.txt
done
