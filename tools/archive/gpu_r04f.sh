#!/bin/bash
# round 4, sixth session: bench contract with the exchange fraction, side-stream test, rank of eight after the publish change
set -u
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_bench_contract.py -q -m gpu -x > $O/benchc.log 2>&1; echo "benchc rc=$?"; tail -4 $O/benchc.log
timeout 600 python -m pytest tests/test_distributed.py -q -m gpu -k "side_stream or parallelograms or one_launch" > $O/dist.log 2>&1; echo "dist rc=$?"; tail -3 $O/dist.log
rb() { timeout 300 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/rank.txt; }
rb --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --world 8 --rank 0 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --world 8 --rank 3 --every 1 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 0 --graph-mode full --steps 1920
rb --world 16 --rank 7 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
cut -c1-20,230- $O/rank.txt
for nx in 125 250 354; do for fl in 0 1; do
  THETIS_AMD_FLOW=$fl timeout 300 python tools/kbench.py --nx $nx --ny $((nx/2)) --steps 384 --prewarm 0.5 --tag flow$fl 2>&1 | tail -1 | cut -c1-160
done; done
THETIS_AMD_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29591 bench.py --gpus 2 --steps 50 --warmup 5 2>/dev/null | grep '^{"metric"' > $O/bench2.json; python -c "
import json; d=json.load(open('$O/bench2.json')); c=d['config']; print(d['value'], d['ms_per_step'], c['exchange'], c['exchange_every'], c['flow'], c['exchange_time_fraction'], c['exchange_time_fraction_note'])"
