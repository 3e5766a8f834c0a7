#!/bin/bash
set -u
O=gpurun_out/r02c; mkdir -p $O
export TMPDIR=/tmp
THETIS_AMD_LIB=$PWD/variants/wt.so timeout 300 python tools/wavetiming.py --nx 125 --ny 500 --stage 1 2>/dev/null > $O/wt_125k.json
THETIS_AMD_LIB=$PWD/variants/wt.so THETIS_AMD_STAGGER=5 timeout 300 python tools/wavetiming.py --nx 125 --ny 500 --stage 1 2>/dev/null > $O/wt_125k_stagger5.json
THETIS_AMD_LIB=$PWD/variants/wt.so THETIS_AMD_STAGE_LDS=20480 timeout 300 python tools/wavetiming.py --nx 125 --ny 500 --stage 1 2>/dev/null > $O/wt_125k_lds8.json
for nx in 125 137 250; do
for cfg in "0 0" "20480 0" "18000 0" "16384 0" "0 2" "0 4" "0 6" "0 8" "0 10" "20480 4" "20480 6" "18000 5"; do
  set -- $cfg
  THETIS_AMD_STAGE_LDS=$1 THETIS_AMD_STAGGER=$2 timeout 300 python tools/kbench.py --nx $nx --ny 500 --tag "lds$1_stg$2" 2>/dev/null | tail -1 >> $O/kbench.log
done
done
cat $O/kbench.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['n_cells'], d['tag'], round(d['us_per_step'],2), round(d['us_per_launch'],2))
"
python - <<'PY'
import json
for f in ('wt_125k','wt_125k_stagger5','wt_125k_lds8'):
    d=json.load(open('gpurun_out/r02c/%s.json'%f)); print(f, json.dumps(d['runs'][-1]))
PY
