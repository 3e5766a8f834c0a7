#!/bin/bash
set -u
O=gpurun_out/r02p; mkdir -p $O
timeout 1800 python -m pytest tests/test_wetting_drying.py tests/test_quads.py tests/test_gpu_fuzz.py tests/test_gpu_examples.py -m gpu -x -q 2>&1 | tail -15
timeout 600 python tools/cfgbench.py 2>/dev/null | grep -E "cfg5|cfg2 tri" 
