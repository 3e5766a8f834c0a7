#!/bin/bash
# round 4, eighth session: the unchanged example script at bench size on one rank and on two ranks sharing the GPU; side-stream test
set -u
O=gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_distributed.py -q -m gpu -k "side_stream" > $O/dist.log 2>&1; echo "side stream rc=$?"; tail -2 $O/dist.log
( time timeout 900 python examples/channel2d.py --nx 1000 --ny 500 --t-end 100 ) > $O/one.log 2>&1; tail -4 $O/one.log
export THETIS_AMD_DIST_BACKEND=gloo
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29601 examples/channel2d.py --nx 1000 --ny 500 --t-end 100 ) > $O/two.log 2>&1; grep -v Gloo $O/two.log | tail -6
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29602 examples/channel2d.py --nx 1000 --ny 500 --t-end 100 ) > $O/four.log 2>&1; grep -v Gloo $O/four.log | tail -6
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29603 examples/channel2d.py --nx 1000 --ny 500 --t-end 100 ) > $O/eight.log 2>&1; grep -v Gloo $O/eight.log | tail -6
