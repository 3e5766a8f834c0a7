#!/bin/bash
# round 6, run 16: two-ring tiles on an unstructured triangulation - runs of the Hilbert order against bisection leaves
set -u
TAG=r06p
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/unstructured_bench.py --points 500000 --steps 60 --triple > $O/${TAG}_unstructured_triple.txt 2> $O/err.txt; tail -3 $O/err.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06p/r06p_unstructured_triple.txt').read().strip().splitlines()[-1])
print(d['n_cells'], d['mesh_build_s'])
for k,v in d['order'].items(): print('{:75s} {:8.2f} us  {:.3f}  {}'.format(k, v['us_per_step'], v['frac_of_8TBs'], v.get('tiles')))
PY
