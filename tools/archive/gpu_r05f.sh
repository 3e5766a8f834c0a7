#!/bin/bash
# round 5, sixth run: the whole GPU suite after the day's changes (D carried with wetting-drying, six-limb sums, flow kernel in its final
# form, multi-block kernel opt-in), per-rank rows of cfg 4 / cfg 5 strips (VERDICT r04 "next 5": none existed), cfg rows
set -u
O=gpurun_out/r05f; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -8 $O/gpu_tests.log | cut -c1-300
rb() { timeout 400 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/rank.txt; }
rb --case cfg2 --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --case cfg2 --world 8 --rank 0 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --case cfg2 --world 8 --rank 3 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
rb --case cfg4 --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480
rb --case cfg4 --world 8 --rank 3 --every 1 --exchange p2p --graph-mode full --steps 480
rb --case cfg4_tracer_only --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480
rb --case cfg5 --world 8 --rank 3 --every 4 --exchange p2p --nosplit --graph-mode full --steps 960
rb --case cfg5 --world 8 --rank 7 --every 4 --exchange p2p --nosplit --graph-mode full --steps 960
rb --case cfg5 --world 8 --rank 3 --every 2 --exchange p2p --nosplit --graph-mode full --steps 960
rb --case cfg2 --world 4 --rank 1 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
rb --case cfg2 --world 2 --rank 0 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
sed 's/"exchange.*"world"/ world/; s/"overlap.*"n_owned"/ n_owned/; s/"n_send.*"flow"/ flow/' $O/rank.txt
timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" > $O/cfgs.txt; cut -c1-200 $O/cfgs.txt
du -sh $O
