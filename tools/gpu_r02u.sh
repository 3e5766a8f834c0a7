#!/bin/bash
set -u
O=gpurun_out/r02u; mkdir -p $O
for sz in "62 500" "125 500" "250 500" "500 500" "1000 500" "2000 500" "2000 1000" "4000 1000"; do
  set -- $sz
  timeout 300 python tools/kbench.py --nx $1 --ny $2 --tag sweep_auto --prewarm 0.5 2>/dev/null | tail -1 >> $O/kbench_sweep.log
done
for sz in "125 500" "137 500" "250 500" "500 500" "1000 500"; do
  set -- $sz
  for b in 0 1; do
    THETIS_AMD_BND_INLINE=$b timeout 300 python tools/kbench.py --nx $1 --ny $2 --tag "binl$b" --prewarm 0.5 2>/dev/null | tail -1 >> $O/kbench_binl.log
  done
done
cat $O/kbench_sweep.log $O/kbench_binl.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['n_cells'], d['tag'], d['order'], round(d['us_per_step'],2), round(d['frac'],3))
"
for a in "--every 4" "--every 4 --exchange p2p --nosplit" "--every 8 --exchange p2p --nosplit"; do
  timeout 300 python tools/rankbench.py --world 8 --rank 3 $a 2>/dev/null | tail -1 >> $O/rankbench.log
done
for w in 2 4; do timeout 300 python tools/rankbench.py --world $w --rank 1 --every 4 --exchange p2p --nosplit 2>/dev/null | tail -1 >> $O/rankbench.log; done
cat $O/rankbench.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['world'], d['exchange'], 'split' if d['split'] else 'nosplit', 'm', d['every'], round(d['us_per_step'],2))
"
timeout 900 python bench.py --no-cpu 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_beyond_cache'])"
timeout 900 python -m pytest tests/test_gpu_solver2d.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
