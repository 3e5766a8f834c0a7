#!/bin/bash
# round 5, run zl: stages 1 + 2 of a step in one launch by overlapped tiles (csrc/swe2d_fuse.h), device numbering in 16 x 6-quad tiles -
# parity / bench-contract / fuzz / solver / example tests, the bench line, every cfg row, against THETIS_AMD_FUSE12=0 on the same box
set -u
O=gpurun_out/r05zl; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_contract.py tests/test_gpu_fuzz.py tests/test_gpu_solver2d.py tests/test_gpu_examples.py tests/test_unstructured.py tests/test_gpu_chunks.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c1-300
for rep in 1 2; do
  for f in on 0; do
    if [ $f = on ]; then unset THETIS_AMD_FUSE12; else export THETIS_AMD_FUSE12=0; fi
    timeout 300 python bench.py 2>/dev/null | tail -1 | cut -c1-2000 > $O/bench_$f.$rep.json
    grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"frac_beyond_cache": [0-9.]*\|"launches_per_step": [0-9]' $O/bench_$f.$rep.json | paste - - - - | sed "s/^/fuse=$f /"
  done
done
unset THETIS_AMD_FUSE12
timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" > $O/cfgs.txt; sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/cfgs.txt | cut -c1-170
for sz in "500 250" "707 354" "1000 500" "1414 707" "2000 1000"; do
  set -- $sz
  for f in on 0; do
    if [ $f = on ]; then unset THETIS_AMD_FUSE12; else export THETIS_AMD_FUSE12=0; fi
    THETIS_AMD_FLOW=0 timeout 300 python tools/kbench.py --nx $1 --ny $2 --steps 60 --prewarm 0.5 2>&1 | grep "^{" | sed "s/^/fuse=$f $1x$2 /" >> $O/kbench_ab.txt
  done
done
unset THETIS_AMD_FUSE12
cut -c1-160 $O/kbench_ab.txt
