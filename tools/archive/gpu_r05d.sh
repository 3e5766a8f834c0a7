#!/bin/bash
# round 5, fourth run: which of the restructured loops of the flow kernel pays (same box, rank 3 of 8 and one device at 125 k cells);
# the consumer of the reference's golden vectors (self-test on oracle vectors); cfg 5 with the 3-wave variant as the default
set -u
O=gpurun_out/r05d; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_reference_golden.py -q -m gpu -rs 2>&1 | tail -6 | cut -c1-400
CFGBENCH_ONLY=cfg5 timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" >> $O/cfg5.txt; cut -c1-260 $O/cfg5.txt
rb() { timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1; }
kb() { THETIS_AMD_FLOW=1 timeout 300 python tools/kbench.py --nx 354 --ny 177 --steps 384 --prewarm 0.5 --tag flow1 2>&1 | tail -1; }
for rep in 1 2; do
  for v in r04 oldall oldpub oldpoll oldrxtx oldpubpoll product; do
    if [ $v = product ]; then rb | sed "s/^/$v /" >> $O/rank_ab.txt; kb | sed "s/^/$v /" >> $O/flow_ab.txt
    else THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so rb | sed "s/^/$v /" >> $O/rank_ab.txt; THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so kb | sed "s/^/$v /" >> $O/flow_ab.txt; fi
  done
done
sed 's/{.*"us_per_step"/ us_per_step/' $O/rank_ab.txt
sed 's/{.*"us_per_step"/ us_per_step/; s/, "us_per_launch.*//' $O/flow_ab.txt
