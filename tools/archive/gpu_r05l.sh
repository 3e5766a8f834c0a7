#!/bin/bash
# round 5, run l: the limiter in one launch (means + boundary-facet means from the last tracer stage): tracer / quad / distributed tests, cfg 4 rows, rank rows of cfg 4
set -u
O=gpurun_out/r05l; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_tracer.py tests/test_quads.py tests/test_distributed.py tests/test_gpu_spmd.py tests/test_gpu_solver2d.py tests/test_gpu_fuzz.py tests/test_gpu_examples.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gpu_tests.log | cut -c1-300
for v in fused unfused; do
  if [ $v = unfused ]; then export THETIS_AMD_LIMITER_UNFUSED=1; else unset THETIS_AMD_LIMITER_UNFUSED; fi
  timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" | grep -i "tracer" | sed "s/^/$v /" >> $O/cfg4.txt
  for c in cfg4 cfg4_tracer_only; do timeout 400 python tools/rankbench.py --case $c --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480 2>&1 | tail -1 | sed "s/^/$v /" >> $O/rank.txt; done
done
unset THETIS_AMD_LIMITER_UNFUSED
cut -c1-200 $O/cfg4.txt; sed 's/"exchange.*"world"/ world/; s/"overlap.*"n_owned"/ n_owned/; s/"n_send.*"flow"/ flow/' $O/rank.txt
