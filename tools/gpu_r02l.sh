#!/bin/bash
set -u
O=gpurun_out/r02l; mkdir -p $O
echo "nproc $(nproc)  cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  affinity $(python -c 'import os; print(len(os.sched_getaffinity(0)))')"
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)" | head -8
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_wetting_drying.py tests/test_distributed.py -m gpu -x -q 2>&1 | tail -3
for nx in 125 137 250 500 1000 2000 4000; do
  timeout 300 python tools/kbench.py --nx $nx --ny 500 --tag "binl166" 2>/dev/null | tail -1 >> $O/kbench.log
done
cat $O/kbench.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['n_cells'], d['tag'], round(d['us_per_step'],2), round(d['us_per_launch'],2), round(d['frac'],3))
"
timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 4 --exchange p2p --nosplit 2>/dev/null | tail -1
timeout 600 python tools/cfgbench.py 2>/dev/null | tail -12
