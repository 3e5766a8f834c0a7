#!/bin/bash
# Round 4, session y: the interior-facet fluxes as one shared function (swe_facet_flux) with jump / average from one interpolation each:
# tests, then A/B against the library before (build_ab_old.so) on one box
set -u
O=gpurun_out/r04y; mkdir -p $O; rm -f $O/*.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_flow_kernel.py tests/test_gpu_fuzz.py tests/test_wetting_drying.py tests/test_gpu_sipg.py tests/test_gpu_solver2d.py tests/test_gpu_tracer.py tests/test_unstructured.py tests/test_distributed.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for rep in 1 2; do
for tag in old new; do
  lib=""; [ $tag = old ] && lib=$R/build_ab_old.so
  for size in "354 177" "500 250" "707 354" "1000 500"; do set -- $size
    THETIS_AMD_LIB=$lib python tools/kbench.py --nx $1 --ny $2 --steps 96 --prewarm 0.4 --tag $tag 2>&1 | grep "^{" | cut -c1-150 >> $O/ab_kbench.txt
  done
  THETIS_AMD_LIB=$lib CFGBENCH_ONLY=cfg5 timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/$tag /" | cut -c1-120 >> $O/ab_cfg5.txt
  THETIS_AMD_LIB=$lib timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1 | sed "s/^/$tag /" | cut -c1-30,200-400 >> $O/ab_rank.txt
done
done
cat $O/ab_kbench.txt $O/ab_cfg5.txt $O/ab_rank.txt
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/pmc_new/pass1 -- python $R/tools/kbench.py --nx 707 --ny 354 --steps 4 > $R/$O/pmc_new.log 2>&1 || echo "pmc failed"
cd $R
python tools/pmc_summary.py $O/pmc_new swe_ > $O/pmc_new.txt
rm -rf $O/pmc_new
grep -E "swe_stage|SQ_INSTS_VALU |SQ_WAVES" $O/pmc_new.txt
