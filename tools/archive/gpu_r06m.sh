#!/bin/bash
# round 6, run 13: the three-stage kernel on 11 x 8-quad patches - from which size, with source terms, in coupled steps
set -u
TAG=r06m
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp THETIS_AMD_TRIPLE_MIN=100000 THETIS_AMD_FLOW=0
kb() { timeout 300 python tools/kbench.py "$@" 2>&1 | tail -1 >> $O/${TAG}_triple_sizes.txt; }
for sz in "388 194" "448 224" "500 250" "592 296" "707 354"; do
  set -- $sz
  for f in 0 1 3 0 1 3; do
    THETIS_AMD_FUSE12=$f kb --nx $1 --ny $2 --steps 40 --prewarm 0.5 --tag fuse$f
  done
done
sed 's/"order.*"n_cells"/"n_cells"/; s/"vol".*//; s/"fused_pair.*"fused_triple"/"fused_triple"/' $O/${TAG}_triple_sizes.txt | cut -c1-200
for f in 1 3 1 3; do
  echo "--- THETIS_AMD_FUSE12=$f" >> $O/${TAG}_cfgs.txt
  CFGBENCH_ONLY=tracers THETIS_AMD_FUSE12=$f timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | head -3 >> $O/${TAG}_cfgs.txt
done
sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/${TAG}_cfgs.txt | cut -c1-200
