#!/bin/bash
mkdir -p gpurun_out/r04o
{
for size in "500 250" "707 354" "1000 500" "2000 1000"; do
  set -- $size
  for lead in 1 2 50; do
    THETIS_AMD_CHAIN_LEAD=$lead python tools/chainbench.py --nx $1 --ny $2
  done
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r04o/chainbench.txt
cat gpurun_out/r04o/chainbench.txt
