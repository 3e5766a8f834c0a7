#!/bin/bash
# A/B of the quadrilateral stage kernels: one lane per cell (THETIS_AMD_QUAD_LANES=1) against two lanes per cell (default)
O=gpurun_out/quad2; mkdir -p $O
python -m pytest tests/test_quads.py tests/test_gpu_sipg.py tests/test_wetting_drying.py -q -m gpu -x 2>&1 | tail -5 > $O/tests.txt
for i in 1 2; do
  THETIS_AMD_QUAD_LANES=1 CFGBENCH_ONLY=quads python tools/cfgbench.py > $O/one_lane_$i.txt 2>&1
  CFGBENCH_ONLY=quads python tools/cfgbench.py > $O/two_lanes_$i.txt 2>&1
done
