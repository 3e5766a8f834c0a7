#!/bin/bash
set -u
O=gpurun_out/r02f; mkdir -p $O
export TMPDIR=/tmp
for nx in 16 31 62 125 250 500; do
THETIS_AMD_LIB=$PWD/variants/wt.so timeout 300 python tools/wavetiming.py --nx $nx --ny 500 --stage 1 2>/dev/null > $O/wt_$nx.json
python - <<PY
import json
d=json.load(open('$O/wt_$nx.json')); r=d['runs'][-1]
print(d['n_cells'], 'span', r['kernel_span_us'], 'wave mean', round(r['wave_total_us']['mean'],2), 'max', r['wave_total_max_us'], 'idx', round(r['index_loads_us'],2), 'loads', round(r['gathers_and_own_loads_us'],2), 'arith', round(r['arithmetic_us'],2), 'st', round(r['stores_us'],2), 'simd hist', r['waves_per_simd_hist'])
print('   tail', json.dumps(r['slowest_5pct']))
PY
done
