#!/bin/bash
# A/B of flow-kernel build variants (variants/*.so via THETIS_AMD_LIB): small meshes through swe2d_advance, one rank of eight
O=gpurun_out/r03b; mkdir -p $O
run() {  # tag lib
  for nx in 125 250 354; do
    THETIS_AMD_LIB=$2 THETIS_AMD_FLOW=1 THETIS_AMD_FUSED_STEP=0 timeout 300 python tools/kbench.py --nx $nx --ny $((nx/2)) --steps 96 --tag $1 2>&1 | tail -1 >> $O/ab.log
  done
  THETIS_AMD_LIB=$2 timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --steps 240 2>&1 | tail -1 | sed "s/^/$1 /" >> $O/ab.log
}
THETIS_AMD_FLOW=0 THETIS_AMD_FUSED_STEP=0 timeout 300 python tools/kbench.py --nx 125 --ny 62 --steps 96 --tag stages 2>&1 | tail -1 >> $O/ab.log
THETIS_AMD_FLOW=0 THETIS_AMD_FUSED_STEP=0 timeout 300 python tools/kbench.py --nx 250 --ny 125 --steps 96 --tag stages 2>&1 | tail -1 >> $O/ab.log
THETIS_AMD_FLOW=0 THETIS_AMD_FUSED_STEP=0 timeout 300 python tools/kbench.py --nx 354 --ny 177 --steps 96 --tag stages 2>&1 | tail -1 >> $O/ab.log
run default $PWD/thetis_amd/libswe2d_hip.so
for v in "$@"; do run $v $PWD/variants/$v.so; done
cat $O/ab.log
