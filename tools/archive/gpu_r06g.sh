#!/bin/bash
# round 6, seventh run: quadrilaterals - the stage arithmetic as one shared function (all quadrilateral / wetting-drying / SIPG / fuzz
# tests), the fused stage pair on quadrilaterals (bitwise test; rows with and without it, 1 M and 640 k cells)
set -u
TAG=r06g
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest -q -m gpu tests/test_quads.py tests/test_gpu_fuzz.py tests/test_gpu_sipg.py tests/test_wetting_drying.py tests/test_gpu_parity.py > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -20 | cut -c1-250
for f in auto 0; do
  if [ $f = auto ]; then unset THETIS_AMD_FUSE12; else export THETIS_AMD_FUSE12=$f; fi
  CFGBENCH_ONLY=quads timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^{/{\"fuse\": \"$f\", /" >> $O/${TAG}_quads.txt
  CFGBENCH_ONLY=quads CFGBENCH_QUAD_N=800 timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^{/{\"fuse\": \"$f\", /" >> $O/${TAG}_quads_640k.txt
done
unset THETIS_AMD_FUSE12
sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/${TAG}_quads.txt | cut -c1-230; echo "--- 640 k cells"; sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/${TAG}_quads_640k.txt | cut -c1-230
du -sh $O
