#!/bin/bash
# round 5, run y: the flow kernel on an unstructured triangulation (124 k triangles) with bisection-box blocks (four-load polling
# instance) against 64 consecutive cells of the Hilbert curve (six loads) and against stage launches; flow tests incl. the
# unstructured cases; rank rows once more
set -u
O=gpurun_out/r05y; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_flow_kernel.py tests/test_unstructured.py tests/test_gpu_fuzz.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c1-300
for rep in 1 2; do
  timeout 600 python tools/unstructured_flow.py --tag boxes 2>&1 | tail -1 >> $O/unstructured.txt
  THETIS_AMD_FLOW_BLOCKS=0 timeout 600 python tools/unstructured_flow.py --tag hilbert 2>&1 | tail -1 >> $O/unstructured.txt
  THETIS_AMD_FLOW=0 timeout 600 python tools/unstructured_flow.py --tag stage_launches 2>&1 | tail -1 >> $O/unstructured.txt
done
cat $O/unstructured.txt
rb() { timeout 300 python tools/rankbench.py --case cfg2 --world 8 --rank $1 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1; }
rb 3 >> $O/rank.txt; rb 0 >> $O/rank.txt
sed 's/"exchange.*"rank"/ "rank"/; s/"every.*"us_per_step"/ us_per_step/' $O/rank.txt
