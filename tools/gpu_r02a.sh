#!/bin/bash
# round-2 GPU session A: distributed tests, p2p timing in loopback, store-policy A/B, kernel timeline of a 125k-cell rank
set -u
O=gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_distributed.py tests/test_gpu_bench_contract.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for a in "--every 4" "--every 4 --exchange p2p" "--every 4 --exchange p2p --nosplit" "--every 8 --exchange p2p --nosplit" "--every 2 --exchange p2p --nosplit" "--every 4 --exchange p2p --nosplit --graph-mode full"; do
  timeout 300 python tools/rankbench.py --world 8 --rank 3 $a 2>&1 | tail -1 >> $O/rankbench.log
done
cat $O/rankbench.log
for lib in default variants/st_sc1.so variants/st_nt.so; do
  for sz in "--nx 125 --ny 500" ""; do
    if [ $lib = default ]; then timeout 300 python tools/kbench.py $sz --tag $lib 2>&1 | tail -1 >> $O/kbench.log
    else THETIS_AMD_LIB=$PWD/$lib timeout 300 python tools/kbench.py $sz --tag $lib 2>&1 | tail -1 >> $O/kbench.log; fi
  done
done
cat $O/kbench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/rankbench.py --world 8 --rank 3 --every 4 --exchange p2p --nosplit --steps 96 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py "$O/prof/*/*kernel_trace.csv" --last 400 > $O/timeline.json 2>&1; cat $O/timeline.json
find $O/prof -name "*.csv" -size +5M -delete
