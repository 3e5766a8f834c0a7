#!/bin/bash
# Round 4, session n: swe2d_advance as two chains of half-launches - bits, then us per step against single launches
mkdir -p gpurun_out/r04n
timeout 900 python -m pytest tests/test_gpu_chains.py -x -q 2>&1 | tail -15 > gpurun_out/r04n/tests.txt
cat gpurun_out/r04n/tests.txt
{
for size in "500 250" "707 354" "1000 500" "1414 707" "2000 1000"; do
  set -- $size
  for lead in 1 2 3 50; do
    THETIS_AMD_CHAIN_LEAD=$lead python tools/kbench.py --nx $1 --ny $2 --steps 96 --prewarm 0.5 --tag "chains lead $lead"
  done
  THETIS_AMD_CHAINS=0 python tools/kbench.py --nx $1 --ny $2 --steps 96 --prewarm 0.5 --tag "single launches, chain numbering"
  THETIS_AMD_CHAIN_MIN_CELLS=100000000 python tools/kbench.py --nx $1 --ny $2 --steps 96 --prewarm 0.5 --tag "single launches, plain numbering"
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r04n/kbench.txt
cat gpurun_out/r04n/kbench.txt
