#!/bin/bash
# round 4, ninth session: after the fix of the flow decision - SPMD suite, the example at bench size under 8 ranks
set -u
O=gpurun_out/r04i; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_spmd.py -q -m gpu > $O/spmd.log 2>&1; echo "spmd rc=$?"; tail -3 $O/spmd.log
export THETIS_AMD_DIST_BACKEND=gloo THETIS_AMD_DIST_TIMEOUT_S=120
( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29603 examples/channel2d.py --nx 1000 --ny 500 --t-end 100 ) > $O/eight.log 2>&1; grep -v "Gloo\|amdgpu.ids\|socket.cpp" $O/eight.log | tail -8 | cut -c1-200
( time timeout 600 python examples/channel2d.py --nx 1000 --ny 500 --t-end 100 ) > $O/one.log 2>&1; tail -5 $O/one.log
