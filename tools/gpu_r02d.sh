#!/bin/bash
set -u
O=gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for nx in 125 137 250 500 1000; do
for b in 0 1; do
  THETIS_AMD_BND_INLINE=$b timeout 300 python tools/kbench.py --nx $nx --ny 500 --tag "binl$b" 2>/dev/null | tail -1 >> $O/kbench.log
done
done
cat $O/kbench.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['n_cells'], d['tag'], round(d['us_per_step'],2), round(d['us_per_launch'],2))
"
THETIS_AMD_BND_INLINE=0 THETIS_AMD_LIB=$PWD/variants/wt.so timeout 300 python tools/wavetiming.py --nx 125 --ny 500 --stage 1 2>/dev/null > $O/wt_125k_b0.json
THETIS_AMD_BND_INLINE=1 THETIS_AMD_LIB=$PWD/variants/wt.so timeout 300 python tools/wavetiming.py --nx 125 --ny 500 --stage 1 2>/dev/null > $O/wt_125k_b1.json
python - <<'PY'
import json
for f in ('wt_125k_b0','wt_125k_b1'):
    d=json.load(open('gpurun_out/r02d/%s.json'%f)); print(f, json.dumps(d['runs'][-1]))
PY
for a in "--every 4 --exchange p2p --nosplit" "--every 4 --exchange p2p"; do
  timeout 300 python tools/rankbench.py --world 8 --rank 3 $a 2>/dev/null | tail -1 >> $O/rankbench.log
done
cat $O/rankbench.log
