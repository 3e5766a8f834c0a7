#!/bin/bash
# round 5, last run: the whole GPU suite, the bench line, smoke(), all cfg rows and the rank rows with the library as committed
set -u
O=gpurun_out/r05_final; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-500
timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" > $O/cfgs.txt; sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/cfgs.txt | cut -c1-170
rb() { timeout 400 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/rank.txt; }
rb --case cfg2 --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --case cfg2 --world 8 --rank 0 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --case cfg5 --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 960
rb --case cfg4 --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480
rb --case cfg4_tracer_only --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480
sed 's/"exchange.*"rank"/ "rank"/; s/"every.*"us_per_step"/ us_per_step/' $O/rank.txt
