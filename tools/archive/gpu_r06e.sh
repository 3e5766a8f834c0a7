#!/bin/bash
# round 6, fifth run: the fused tracer step (bitwise), one exchange launch pair per coupled cycle on partitions, cfg 4 on the
# 4 M-triangle channel with and without the three-stage kernels
set -u
TAG=r06e
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest -q -m gpu tests/test_gpu_tracer.py -k fused tests/test_distributed.py -k "tracer or coupled" tests/test_gpu_spmd.py -k "tracer or example or user_script" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -10 | cut -c1-250
rb() { timeout 400 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/${TAG}_rank.txt; }
rb --case cfg4 --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480
rb --case cfg4_tracer_only --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480
rb --case cfg4 --world 8 --rank 3 --every 1 --exchange p2p --graph-mode full --steps 480
rb --case cfg4 --world 8 --rank 3 --every 4 --exchange p2p --graph-mode full --steps 480
rb --case cfg4 --world 2 --rank 0 --every 2 --exchange p2p --graph-mode full --steps 240
python - <<'PY'
import json
for l in open('gpurun_out/r06e/r06e_rank.txt'):
    try: d = json.loads(l)
    except Exception: print(l[:200]); continue
    print(d['case'], 'world', d['world'], 'every', d['every'], 'ghost', d['n_ghost'], 'fused', d['fused_pair'][:1], 'us/step %.2f' % d['us_per_step'])
PY
for f in auto 2; do
  if [ $f = auto ]; then unset THETIS_AMD_FUSE12; else export THETIS_AMD_FUSE12=$f; fi
  CFGBENCH_ONLY=tracers CFGBENCH_NX=2000 timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" | sed "s/^{/{\"fuse\": \"$f\", /" >> $O/${TAG}_cfgs_4m.txt
done
unset THETIS_AMD_FUSE12
sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/${TAG}_cfgs_4m.txt | cut -c1-230
du -sh $O
