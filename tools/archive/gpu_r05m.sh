#!/bin/bash
# round 5, run m: the library compiled with -mllvm -disable-machine-licm (no scratch in any flow kernel variant, 233-240 instead of
# 254-256 VGPRs; triangle stage kernels 116-158 instead of 134-166, the first-stage epilogue variant at 128 = four waves per SIMD;
# all quadrilateral kernels <= 168) against the product build: flow kernel tests, one device at several sizes with both boundary
# variants, rank 3 of 8, all cfg rows
set -u
O=gpurun_out/r05m; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_nolicm.so timeout 1500 python -m pytest tests/test_gpu_flow_kernel.py tests/test_gpu_parity.py tests/test_quads.py -q -m gpu -x > $O/gpu_tests_nolicm.log 2>&1; echo "gpu tests (nolicm) rc=$?"; tail -3 $O/gpu_tests_nolicm.log | cut -c1-300
rb() { timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1; }
kb() { timeout 300 python tools/kbench.py --nx $1 --ny $2 --steps $3 --prewarm 0.5 --tag $4 2>&1 | tail -1; }
for rep in 1 2; do
  for v in product nolicm; do
    if [ $v = nolicm ]; then export THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_nolicm.so; else unset THETIS_AMD_LIB; fi
    rb | sed "s/^/$v /" >> $O/rank_ab.txt
    THETIS_AMD_FLOW=1 kb 354 177 384 flow1 | sed "s/^/$v /" >> $O/flow_ab.txt
    for binl in 1 0; do
      THETIS_AMD_BND_INLINE=$binl THETIS_AMD_FLOW=0 kb 1000 500 100 binl$binl | sed "s/^/$v /" >> $O/stage_ab.txt
      THETIS_AMD_BND_INLINE=$binl THETIS_AMD_FLOW=0 kb 708 354 150 binl$binl | sed "s/^/$v /" >> $O/stage_ab.txt
      THETIS_AMD_BND_INLINE=$binl THETIS_AMD_FLOW=0 kb 354 177 300 binl$binl | sed "s/^/$v /" >> $O/stage_ab.txt
    done
  done
done
unset THETIS_AMD_LIB
sed 's/{.*"us_per_step"/ us_per_step/' $O/rank_ab.txt
sed 's/"order.*"n_cells"/"n_cells"/; s/, "us_per_launch.*//' $O/flow_ab.txt $O/stage_ab.txt
for v in product nolicm; do
  if [ $v = nolicm ]; then export THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_nolicm.so; else unset THETIS_AMD_LIB; fi
  timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/$v /" >> $O/cfgs.txt
done
unset THETIS_AMD_LIB
sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/cfgs.txt | cut -c1-170
