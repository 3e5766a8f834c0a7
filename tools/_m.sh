python -m pytest tests/test_gpu_flow_kernel.py -q -x 2>&1 | tail -2
python -m pytest tests/test_distributed.py -q -x -m gpu -k "flow or one_launch" 2>&1 | tail -2
for rep in 1 2 3; do for rank in 3 0; do
  timeout 300 python tools/rankbench.py --world 8 --rank $rank --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('rank$rank fx  %.2f us/step  timeouts %d' % (d['us_per_step'], d['flow_timeouts']))"
done; done
timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 1 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240 2>&1 | tail -1 | cut -c300-
timeout 300 python tools/rankbench.py --world 16 --rank 7 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240 2>&1 | tail -1 | cut -c300-
timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 0 --graph-mode full --steps 240 2>&1 | tail -1 | cut -c300-
