#!/bin/bash
# round 6, tenth run: the in-launch exchange with one exchange per step on four ranks (first contact's dropped candidate)
set -u
TAG=r06j
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_distributed.py -q -m gpu -k "one_launch_per_cycle and (channel64 or channel256)" > $O/tests.log 2>&1; echo "tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/tests.log | tail -30 | cut -c1-400
du -sh $O
