#!/bin/bash
set -u
O=gpurun_out/r02g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for nx in 125 137 250 500 1000; do
for b in 0 1; do
  THETIS_AMD_BND_INLINE=$b timeout 300 python tools/kbench.py --nx $nx --ny 500 --tag "binl$b" 2>/dev/null | tail -1 >> $O/kbench.log
done
THETIS_AMD_LIB=$PWD/variants/minw3.so THETIS_AMD_BND_INLINE=1 timeout 300 python tools/kbench.py --nx $nx --ny 500 --tag "minw3_binl1" 2>/dev/null | tail -1 >> $O/kbench.log
done
cat $O/kbench.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['n_cells'], d['tag'], round(d['us_per_step'],2), round(d['us_per_launch'],2))
"
for nx in 125; do
THETIS_AMD_LIB=$PWD/variants/wt.so timeout 300 python tools/wavetiming.py --nx $nx --ny 500 --stage 1 2>/dev/null > $O/wt_$nx.json
python - <<PY
import json
d=json.load(open('$O/wt_$nx.json')); r=d['runs'][-1]
print(d['n_cells'], 'span', r['kernel_span_us'], 'wave mean', round(r['wave_total_us']['mean'],2), 'max', r['wave_total_max_us'], 'idx', round(r['index_loads_us'],2), 'loads', round(r['gathers_and_own_loads_us'],2), 'arith', round(r['arithmetic_us'],2), 'st', round(r['stores_us'],2), 'simd hist', r['waves_per_simd_hist'])
print('   tail', json.dumps(r['slowest_5pct']))
PY
done
timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 4 --exchange p2p --nosplit 2>/dev/null | tail -1
