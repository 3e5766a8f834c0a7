#!/bin/bash
O=gpurun_out/r03j; mkdir -p $O
for alt in 0 1; do
  echo "== THETIS_AMD_ALTERNATE=$alt" >> $O/cfg.log
  THETIS_AMD_ALTERNATE=$alt timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-60s %8d cells %8.1f us/step  frac %.3f' % (d['config'][:60], d['n_cells'], d['us_per_step'], d['frac_of_8TBs']))
" >> $O/cfg.log
done
cat $O/cfg.log
timeout 600 python bench.py --no-cpu > $O/bench.json 2> $O/bench.err; python -c "
import json; d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_beyond_cache'])"
