#!/bin/bash
# Round 4, session y2: the quadrilateral kernel takes the shared facet-flux function too: tests, A/B of the quadrilateral rows
set -u
O=gpurun_out/r04y2; mkdir -p $O; rm -f $O/*.txt
R=$GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_quads.py tests/test_gpu_tracer.py tests/test_gpu_sipg.py tests/test_wetting_drying.py tests/test_gpu_solver2d.py tests/test_gpu_fuzz.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 900 python -m pytest tests/test_distributed.py tests/test_gpu_spmd.py -q -m gpu -x -k "quad" > $O/tests2.log 2>&1; echo "tests2 rc=$?"; tail -2 $O/tests2.log
for rep in 1 2; do
for tag in old new; do
  lib=""; [ $tag = old ] && lib=$R/build_ab_old.so
  THETIS_AMD_LIB=$lib CFGBENCH_ONLY=quads timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/$tag /" | cut -c1-170 >> $O/ab_quads.txt
done
done
cat $O/ab_quads.txt
