#!/bin/bash
set -u
timeout 1500 python -m pytest tests/test_quads.py tests/test_gpu_fuzz.py tests/test_gpu_sipg.py tests/test_wetting_drying.py tests/test_meshio.py tests/test_distributed.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python tools/cfgbench.py 2>/dev/null | grep -E "quadr|cfg5"
THETIS_AMD_LDSX=0 timeout 600 python tools/cfgbench.py 2>/dev/null | grep -E "quadrilaterals SWE"
