#!/bin/bash
# round 6, eighth run: the memory-safety pass (-DSWE_RANGE_CHECK incl. the fused kernels, negative control) and the adversary builds
# of the granule protocol (tools/range_check.sh), then the whole GPU suite with the product library
set -u
TAG=r06h
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
bash tools/range_check.sh > $O/${TAG}_range_check.txt 2>&1; grep -E "range check:|negative control|passed|failed|rc=" $O/${TAG}_range_check.txt | cut -c1-200
timeout 3000 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | tail -20 | cut -c1-220
du -sh $O
