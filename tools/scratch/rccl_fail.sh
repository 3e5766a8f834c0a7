#!/bin/bash
# first-contact rehearsal: two ranks on ONE GPU with the real nccl backend - RCCL refuses the duplicate device; the bench must
# record the failure, keep the peer-to-peer transport and print its JSON line
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 WORLD_SIZE=2 LOCAL_RANK=0 THETIS_AMD_DIST_TIMEOUT_S=60
RANK=1 timeout 900 python bench.py --gpus 2 --steps 16 --warmup 2 --prewarm 0.05 > /tmp/r1.log 2>&1 &
P=$!
RANK=0 timeout 900 python bench.py --gpus 2 --steps 16 --warmup 2 --prewarm 0.05 > /tmp/r0.log 2>&1
echo "rank0 rc=$?"; wait $P; echo "rank1 rc=$?"
grep -E "thetis_amd\]" /tmp/r0.log | cut -c1-300 | head -10
tail -1 /tmp/r0.log | cut -c1-1500
