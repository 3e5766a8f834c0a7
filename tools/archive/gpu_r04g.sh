#!/bin/bash
# round 4, seventh session: quadrilateral connectivity packed into two 16-byte records - A/B on one box, tests
set -u
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  for lib in build_dbg/base_quads.so thetis_amd/libswe2d_hip.so; do
    echo "lib $lib" >> $O/quads_ab.txt
    THETIS_AMD_LIB=$PWD/$lib CFGBENCH_ONLY=quads timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" | cut -c1-200 >> $O/quads_ab.txt
  done
done
cat $O/quads_ab.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('lib'): print(l.strip()); continue
    d = json.loads(l[:l.rindex('}')+1]) if l.strip().endswith('}') else None
    if d: print('   %-70s %7.1f us  %.3f' % (d['config'], d['us_per_step'], d['frac_of_8TBs']))
"
timeout 1200 python -m pytest tests/test_quads.py tests/test_gpu_tracer.py tests/test_gpu_sipg.py tests/test_wetting_drying.py -q -m gpu > $O/quadtests.log 2>&1; echo "quad tests rc=$?"; tail -3 $O/quadtests.log
timeout 600 python -m pytest tests/test_distributed.py -q -m gpu -k "side_stream or quad" > $O/dist.log 2>&1; echo "dist rc=$?"; tail -3 $O/dist.log
