#!/bin/bash
# round 5, run k: cfg 5 taken apart (wetting-drying / Manning on and off): us per step and instruction counters per stage kernel
set -u
O=$PWD/gpurun_out/r05k; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
CFGBENCH_ONLY=cfg5_parts timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" > $O/parts.txt; cut -c1-220 $O/parts.txt
cd /tmp
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES"; do
  CFGBENCH_ONLY=cfg5_parts_profile timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pass3 -- python $R/tools/cfgbench.py > $O/pass3.log 2>&1 || echo "pass failed"
done
cd $R
python tools/pmc_summary.py $O swe_stage_kernel > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt | cut -c1-200
find $O/pass3 -name "*.csv" -size +2000k -delete; du -sh $O
