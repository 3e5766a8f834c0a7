#!/bin/bash
# round 6, ninth run: the three tests run 8 left red (quadrilateral rule at 850 k cells; first contact's story with a failure line)
set -u
TAG=r06i
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_quads.py tests/test_gpu_bench_contract.py -q -m gpu -k "fused_stage_pair_on_quadrilaterals or first_contact" > $O/tests.log 2>&1; echo "tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|FAILED / dropped|^E  " $O/tests.log | tail -30 | cut -c1-400
THETIS_AMD_LARGE_MESH=512,64 THETIS_AMD_SETUP_BUDGET_S=25 THETIS_AMD_SOAK_S=0.3 timeout 600 python -m tools.first_contact --gpus 4 --same-gpu --mesh 256,64 --steps 16 --warmup 2 --port 29611 --log $O/first_contact.log > /dev/null 2> $O/first_contact.err
grep -n "FAILED\|failures\|  - " $O/first_contact.log | cut -c1-300
du -sh $O
