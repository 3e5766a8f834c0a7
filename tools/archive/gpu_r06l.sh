#!/bin/bash
# round 6, run 12: the two-ring tiles of the three-stage kernel as patches of their own (swe2d_fused_set_triple_tiles) - which patch?
set -u
TAG=r06l
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest -q -m gpu tests/test_gpu_parity.py -k "triple or three_stages" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/tests.log | tail -12 | cut -c1-250
kb() { timeout 300 python tools/kbench.py "$@" 2>&1 | tail -1 >> $O/${TAG}_triple_tiles.txt; }
for sz in "707 354" "1000 500" "1414 707" "2000 1000"; do
  set -- $sz
  THETIS_AMD_FUSE12=1 kb --nx $1 --ny $2 --steps 40 --prewarm 0.5 --tag pair
  for t in 0 12,7 16,5 14,6 11,8 10,8; do
    THETIS_AMD_FUSE12=3 THETIS_AMD_TRIPLE_TILE=$t kb --nx $1 --ny $2 --steps 40 --prewarm 0.5 --tag triple_$t
  done
  THETIS_AMD_FUSE12=1 kb --nx $1 --ny $2 --steps 40 --prewarm 0.5 --tag pair
done
sed 's/"order.*"n_cells"/"n_cells"/; s/"vol".*//; s/"fused_pair.*"fused_triple"/"fused_triple"/' $O/${TAG}_triple_tiles.txt | cut -c1-220
