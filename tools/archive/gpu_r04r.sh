#!/bin/bash
# Round 4, session r: the arithmetic cuts of the stage kernels as committed (Manning by orbit sums, lean constant-coefficient path,
# third-order x^(-1/3), clamped sums of squares, dry-node test without the quotient): GPU tests that touch them, cfg rows old / new
set -u
O=gpurun_out/r04r; mkdir -p $O; rm -f $O/*.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_wetting_drying.py tests/test_quads.py tests/test_gpu_flow_kernel.py tests/test_gpu_fuzz.py tests/test_gpu_solver2d.py tests/test_gpu_sipg.py tests/test_gpu_tracer.py tests/test_gpu_examples.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
for rep in 1 2; do
for tag in old new; do
  lib=""; [ $tag = old ] && lib=$R/build_ab_old.so
  THETIS_AMD_LIB=$lib CFGBENCH_ONLY=cfg5_parts timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/$tag /" >> $O/ab_cfg5_parts.txt
  THETIS_AMD_LIB=$lib CFGBENCH_ONLY=cfg5 timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/$tag /" >> $O/ab_cfg5.txt
  THETIS_AMD_LIB=$lib CFGBENCH_ONLY=quads timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/$tag /" >> $O/ab_quads.txt
done
done
cut -c1-140 $O/ab_cfg5_parts.txt; cut -c1-190 $O/ab_cfg5.txt; cut -c1-160 $O/ab_quads.txt
cd /tmp
CFGBENCH_ONLY=cfg5_parts_profile rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/pmc_new/pass1 -- python $R/tools/cfgbench.py > $R/$O/pmc_new.log 2>&1 || echo "pmc failed"
cd $R
python tools/pmc_summary.py $O/pmc_new swe_ > $O/pmc_new.txt
rm -rf $O/pmc_new
grep -E "swe_stage|SQ_INSTS_VALU " $O/pmc_new.txt
