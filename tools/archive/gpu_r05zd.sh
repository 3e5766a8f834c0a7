#!/bin/bash
# round 5, run zd: the 16-B connectivity records where they pay (conn_pays: launches of >= 250 k cells of kernels that stream) -
# parity tests (forced in every launch / by the rule / never), then bench line and rows with the rule against THETIS_AMD_COMPACT_IDX=0
set -u
O=gpurun_out/r05zd; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracer.py -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c1-300
for rep in 1 2 3; do
  for c in 1 0; do
    export THETIS_AMD_COMPACT_IDX=$c
    timeout 300 python bench.py 2>/dev/null | tail -1 | cut -c1-420 | sed "s/^/compact=$c /" >> $O/bench_ab.txt
    CFGBENCH_ONLY=tracers timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^/compact=$c /" >> $O/cfg_ab.txt
  done
done
unset THETIS_AMD_COMPACT_IDX
sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/cfg_ab.txt | cut -c1-170
grep -o 'compact=.\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $O/bench_ab.txt | paste - - -
