#!/bin/bash
# round 5, run zc: triangle kernels read the connectivity as 16-B records of differences (csrc/swe2d_conn.h) - same-box A/B against
# THETIS_AMD_COMPACT_IDX=0 (the 24-B records) at four launch sizes, the bench line, the tracer rows, cfg 5; the new parity test
set -u
O=gpurun_out/r05zc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "compact_connectivity or variants_agree or alternating" > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c1-300
for rep in 1 2 3; do
  for c in 1 0; do
    export THETIS_AMD_COMPACT_IDX=$c
    for sz in "354 177" "707 354" "1000 500" "2000 1000"; do
      set -- $sz
      THETIS_AMD_FLOW=0 timeout 300 python tools/kbench.py --nx $1 --ny $2 --steps 60 --prewarm 0.5 2>&1 | grep "^{" | sed "s/^/compact=$c $1x$2 /" >> $O/kbench_ab.txt
    done
  done
done
for rep in 1 2; do
  for c in 1 0; do
    export THETIS_AMD_COMPACT_IDX=$c
    timeout 300 python bench.py 2>/dev/null | tail -1 | cut -c1-420 | sed "s/^/compact=$c /" >> $O/bench_ab.txt
    CFGBENCH_ONLY=tracers timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/compact=$c /" >> $O/cfg_ab.txt
    CFGBENCH_ONLY=cfg5 timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/compact=$c /" >> $O/cfg_ab.txt
  done
done
unset THETIS_AMD_COMPACT_IDX
cut -c1-200 $O/kbench_ab.txt
sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/cfg_ab.txt | cut -c1-170
grep -o 'compact=.\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $O/bench_ab.txt | paste - - -
