#!/bin/bash
# round 4, fourth session: the whole GPU suite with the library as committed
set -u
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
timeout 3300 python -m pytest tests -q -m gpu > $O/all.log 2>&1; echo "all rc=$?"; tail -8 $O/all.log
