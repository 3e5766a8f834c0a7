#!/bin/bash
# round 5, third run: wetting-drying with D carried on the device (cfg 5) - the tests that touch it, cfg 5 timing; same-box A/B of the
# rank of eight: round-4 library / this round's with and without s_setprio around the chain segment
set -u
O=gpurun_out/r05c; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_wetting_drying.py tests/test_gpu_sipg.py tests/test_gpu_fuzz.py tests/test_quads.py tests/test_gpu_tracer.py tests/test_gpu_parity.py tests/test_gpu_solver2d.py tests/test_gpu_examples.py -q -m gpu > $O/wd_tests.log 2>&1; echo "wd tests rc=$?"; tail -15 $O/wd_tests.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_spmd.py -q -m gpu -k "balzano or coast or fields" > $O/wd_spmd.log 2>&1; echo "wd spmd rc=$?"; tail -3 $O/wd_spmd.log | cut -c1-300
for b in 1 0; do THETIS_AMD_BND_INLINE=$b CFGBENCH_ONLY=cfg5 timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/BINL=$b /" >> $O/cfg5.txt; done
THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_r04.so CFGBENCH_ONLY=cfg5 timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/r04 library /" >> $O/cfg5.txt
cut -c1-260 $O/cfg5.txt
rb() { timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1; }
for rep in 1 2; do
  for v in r04 noprio product; do
    if [ $v = product ]; then rb | sed "s/^/$v /" >> $O/rank_ab.txt; else THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so rb | sed "s/^/$v /" >> $O/rank_ab.txt; fi
  done
done
sed 's/{.*"us_per_step"/ us_per_step/' $O/rank_ab.txt
for v in r04 noprio product; do
  if [ $v = product ]; then THETIS_AMD_FLOW=1 timeout 300 python tools/kbench.py --nx 354 --ny 177 --steps 384 --prewarm 0.5 --tag flow1 2>&1 | tail -1 | sed "s/^/$v /" >> $O/flow_sizes.txt
  else THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so THETIS_AMD_FLOW=1 timeout 300 python tools/kbench.py --nx 354 --ny 177 --steps 384 --prewarm 0.5 --tag flow1 2>&1 | tail -1 | sed "s/^/$v /" >> $O/flow_sizes.txt; fi
done
cut -c1-200 $O/flow_sizes.txt
du -sh $O
