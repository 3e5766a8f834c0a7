#!/bin/bash
# round 5, run x: per-block time stamps of a stage of the flow kernel after the round's changes (rank 3 of 8, tools/rankbench.py --timing,
# -DSWE_WAVE_TIMING -DSWE_FLOW_TS_STAGE=8 build without machine LICM); the pause between two polling passes (s_sleep 0 / 2 / 6 / 12)
set -u
O=gpurun_out/r05x; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_wt8.so timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 --timing 2>&1 | tail -2 | sed "s/^/stage 8: /" > $O/rank_timing.txt
cut -c1-900 $O/rank_timing.txt
rb() { timeout 300 python tools/rankbench.py --case cfg2 --world 8 --rank $1 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1; }
kb() { THETIS_AMD_FLOW=1 timeout 300 python tools/kbench.py --nx 354 --ny 177 --steps 384 --prewarm 0.5 --tag flow1 2>&1 | tail -1; }
for rep in 1 2; do
  for v in product sleep0 sleep6 sleep12; do
    if [ $v = product ]; then unset THETIS_AMD_LIB; else export THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so; fi
    rb 3 | sed "s/^/$v /" >> $O/rank_ab.txt
    rb 0 | sed "s/^/$v /" >> $O/rank_ab.txt
    kb | sed "s/^/$v /" >> $O/flow_ab.txt
  done
done
unset THETIS_AMD_LIB
sed 's/"exchange.*"rank"/ "rank"/; s/"every.*"us_per_step"/ us_per_step/' $O/rank_ab.txt
sed 's/"order.*"n_cells"/"n_cells"/; s/, "us_per_launch.*//' $O/flow_ab.txt
