#!/bin/bash
# round 5, run t: a polling pass that reads the SIX value granules of every incoming slot (three loads per lane for up to 32 rim
# facets: instances with 3 / 6 / 9 loads) = 'product', against the library of run s (all eight granules of a slot: 4 / 8 / 9 loads) =
# 'poll4'; both on the 8 x 4-quad blocks; flow / distributed / spmd / fuzz tests and the adversary builds
set -u
O=gpurun_out/r05t; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_flow_kernel.py tests/test_distributed.py tests/test_gpu_spmd.py tests/test_gpu_fuzz.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c1-300
rb() { timeout 300 python tools/rankbench.py --case $1 --world 8 --rank $2 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1; }
kb() { THETIS_AMD_FLOW=1 timeout 300 python tools/kbench.py --nx $1 --ny $2 --steps 384 --prewarm 0.5 --tag flow1 2>&1 | tail -1; }
for rep in 1 2 3; do
  for v in product poll4; do
    if [ $v = product ]; then unset THETIS_AMD_LIB; else export THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so; fi
    rb cfg2 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb cfg2 0 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb cfg2_src 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb cfg5 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    kb 354 177 | sed "s/^/$v /" >> $O/flow_ab.txt
  done
done
unset THETIS_AMD_LIB
sed 's/"exchange.*"rank"/ "rank"/; s/"every.*"us_per_step"/ us_per_step/' $O/rank_ab.txt
sed 's/"order.*"n_cells"/"n_cells"/; s/, "us_per_launch.*//' $O/flow_ab.txt
timeout 600 python tools/kbench.py --steps 100 --prewarm 0.5 --tag cfg2_1M 2>&1 | tail -1 | cut -c1-200
