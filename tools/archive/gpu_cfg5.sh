#!/bin/bash
# cfg 5 (wetting-drying + Manning): boundary-inline against epilogue variant, and the PMC counters of the stage kernel
O=$PWD/gpurun_out/cfg5; mkdir -p $O; R=$PWD
for rep in 1 2; do for b in 1 0; do
  echo "THETIS_AMD_BND_INLINE=$b" >> $O/ab.txt
  THETIS_AMD_BND_INLINE=$b CFGBENCH_ONLY=cfg5 python tools/cfgbench.py 2>&1 | grep "^{" >> $O/ab.txt
done; done
CFGBENCH_ONLY=cfg5_profile bash tools/pmc.sh $O/pmc python $R/tools/cfgbench.py
python $R/tools/pmc_summary.py $O/pmc swe_stage_kernel > $O/pmc_summary.txt
rm -rf $O/pmc
