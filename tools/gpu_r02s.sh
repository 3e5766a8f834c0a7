#!/bin/bash
set -u
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_quads.py tests/test_gpu_solver2d.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 4 --exchange p2p --nosplit 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rank3of8', round(d['us_per_step'],2))"; done
for sz in "125 500" "1000 500"; do set -- $sz; timeout 300 python tools/kbench.py --nx $1 --ny $2 --prewarm 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_cells'], round(d['us_per_step'],2))"; done
THETIS_AMD_LIB=$PWD/variants/wt.so timeout 300 python tools/wavetiming.py --nx 125 --ny 500 --stage 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['runs'][-1]; print('span', r['kernel_span_us'], 'mean', round(r['wave_total_us']['mean'],2), 'max', r['wave_total_max_us']); print(r['slowest_5pct'])"
