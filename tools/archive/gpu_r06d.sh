#!/bin/bash
# round 6, fourth run: the tracer kernels after their arithmetic moved into one shared function without implicit contraction, the
# tracer's three stages in one launch (bitwise test, cfg 4 rows with and without it)
set -u
TAG=r06d
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest -q -m gpu tests/test_gpu_tracer.py tests/test_gpu_sipg.py tests/test_quads.py tests/test_gpu_fuzz.py tests/test_unstructured.py tests/test_gpu_examples.py > $O/tests_tracer.log 2>&1; echo "tracer tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_tracer.log | tail -20 | cut -c1-250
timeout 2400 python -m pytest -q -m gpu tests/test_distributed.py -k "tracer or coupled" tests/test_gpu_spmd.py -k "tracer or example" > $O/tests_dist.log 2>&1; echo "dist tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_dist.log | tail -20 | cut -c1-250
CFGBENCH_ONLY=tracers timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" > $O/${TAG}_cfgs.txt
CFGBENCH_ONLY=tracers THETIS_AMD_FUSE12=0 timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" > $O/${TAG}_cfgs_nofuse.txt
sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/${TAG}_cfgs.txt | cut -c1-230; echo "--- without fusion"; sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/${TAG}_cfgs_nofuse.txt | cut -c1-230
du -sh $O
