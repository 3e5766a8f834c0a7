#!/bin/bash
# round 6, run 17: the source-term instances of the three-stage kernel at two workgroups per CU (no scratch) against the fused pair
set -u
TAG=r06q
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu tests/test_gpu_parity.py -k "triple" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/tests.log | tail -8 | cut -c1-250
for f in 2 3 2 3; do
  echo "--- THETIS_AMD_FUSE12=$f" >> $O/${TAG}_cfgs.txt
  CFGBENCH_ONLY=tracers THETIS_AMD_FUSE12=$f timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | head -2 >> $O/${TAG}_cfgs.txt
  CFGBENCH_NX=2000 CFGBENCH_ONLY=tracers THETIS_AMD_FUSE12=$f timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | head -2 >> $O/${TAG}_cfgs.txt
  CFGBENCH_NX=500 CFGBENCH_ONLY=tracers THETIS_AMD_FUSE12=$f timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | head -2 >> $O/${TAG}_cfgs.txt
done
sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/${TAG}_cfgs.txt | cut -c1-200
