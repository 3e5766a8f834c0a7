#!/bin/bash
# round 5, run j: the receive of an exchange cycle inside the flow launch - five instead of three system-scope granule loads per lane
# in flight (one trip instead of two for a block next to a cut), with and without the one-lane hint phase; rank 3 and rank 0 of 8
set -u
O=gpurun_out/r05j; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
rb() { timeout 300 python tools/rankbench.py --world 8 --rank $1 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1; }
for rep in 1 2; do
  for v in product rx5 rx5nohint nohint; do
    for r in 3 0; do
      if [ $v = product ]; then rb $r | sed "s/^/$v rank $r /" >> $O/rank_ab.txt
      else THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so rb $r | sed "s/^/$v rank $r /" >> $O/rank_ab.txt; fi
    done
  done
done
sed 's/{.*"us_per_step"/ us_per_step/' $O/rank_ab.txt
