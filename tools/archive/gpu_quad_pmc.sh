#!/bin/bash
O=$PWD/gpurun_out/quad2; mkdir -p $O
R=$PWD
export THETIS_AMD_QUAD_LANES=1
for nx in 700 1000; do
  bash tools/pmc.sh $O/pmc_$nx python $R/tools/quadrun.py $nx 12
  python $R/tools/pmc_summary.py $O/pmc_$nx swe_stage_kernel_quad > $O/pmc_${nx}_summary.txt
  rm -rf $O/pmc_$nx
done
