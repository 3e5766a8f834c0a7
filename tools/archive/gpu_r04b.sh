#!/bin/bash
# round 4, second GPU session: the multi-rank tests after their fixes, the diagnostics kernel, cfg 5 taken apart (timing + PMC)
set -u
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_spmd.py -q -m gpu > $O/spmd.log 2>&1; echo "spmd rc=$?"; tail -5 $O/spmd.log
timeout 900 python -m pytest tests/test_distributed.py -q -m gpu -k "eight or refused or coupled" > $O/dist8.log 2>&1; echo "dist8 rc=$?"; tail -4 $O/dist8.log
timeout 900 python -m pytest tests/test_gpu_tracer.py tests/test_gpu_solver2d.py tests/test_gpu_flow_kernel.py -q -m gpu > $O/some.log 2>&1; echo "some rc=$?"; tail -3 $O/some.log
CFGBENCH_ONLY=cfg5_parts timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" > $O/cfg5_parts.txt; cat $O/cfg5_parts.txt
CFGBENCH_ONLY=cfg5_parts_profile bash tools/pmc.sh $R/$O/pmc python $R/tools/cfgbench.py
cd $R
python tools/pmc_summary.py $O/pmc swe_ > $O/cfg5_parts_pmc.txt
rm -rf $O/pmc
grep -E "swe_stage|SQ_INSTS_VALU |SQ_WAVES|GRBM_GUI" $O/cfg5_parts_pmc.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -- python $R/examples/channel2d.py --nx 1000 --ny 500 --t-end 50 > $R/$O/kstats.log 2>&1
cd $R
cp $(ls $O/kstats/*/*kernel_stats.csv | head -1) $O/channel_kernel_stats.csv 2>/dev/null; head -8 $O/channel_kernel_stats.csv
find $O -name "*.csv" -size +3M -delete
