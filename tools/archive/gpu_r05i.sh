#!/bin/bash
# round 5, run i: the whole GPU suite with the three-wave quadrilateral kernels; quadrilateral rows at the bench size (beyond the
# Infinity Cache) and at 640 k cells (inside it); all cfg rows; bench line
set -u
O=gpurun_out/r05i; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log | cut -c1-300
for n in 1000 800 700; do CFGBENCH_QUAD_N=$n CFGBENCH_ONLY=quads timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/n=$n /" >> $O/quads_sizes.txt; done
cut -c1-190 $O/quads_sizes.txt
timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" > $O/cfgs.txt; cut -c1-200 $O/cfgs.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-600
