#!/bin/bash
set -u
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_distributed.py tests/test_gpu_bench_contract.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for a in "--every 4" "--every 4 --exchange p2p" "--every 4 --exchange p2p --nosplit" "--every 8 --exchange p2p --nosplit"; do
  timeout 300 python tools/rankbench.py --world 8 --rank 3 $a 2>&1 | tail -1 >> $O/rankbench.log
done
cat $O/rankbench.log
THETIS_AMD_LIB=$PWD/variants/wt.so timeout 300 python tools/wavetiming.py --nx 125 --ny 500 --stage 1 > $O/wt_125k.json 2>&1
THETIS_AMD_LIB=$PWD/variants/wt.so timeout 300 python tools/wavetiming.py --nx 1000 --ny 500 --stage 1 > $O/wt_1m.json 2>&1
python - <<'PY'
import json
for f in ('gpurun_out/r02b/wt_125k.json','gpurun_out/r02b/wt_1m.json'):
    try:
        d=json.load(open(f)); print(f, json.dumps(d['runs'][-1]))
    except Exception as e:
        print(f, 'ERR', e, open(f).read()[-600:])
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/rankbench.py --world 8 --rank 3 --every 4 --exchange p2p --nosplit --steps 96 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py "$O/prof/*/*kernel_trace.csv" --last 400 > $O/timeline.json 2>&1; cat $O/timeline.json
find $O/prof -name "*.csv" -size +5M -delete
