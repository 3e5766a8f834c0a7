#!/bin/bash
# round 4, first GPU session: the new multi-rank tests first, then the whole GPU suite, the bench line and its kernel-trace stats
set -u
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_spmd.py -x -q -m gpu > $O/spmd.log 2>&1; echo "spmd rc=$?"; tail -3 $O/spmd.log
timeout 900 python -m pytest tests/test_distributed.py -x -q -m gpu -k "eight or one_launch or peer_to_peer" > $O/dist8.log 2>&1; echo "dist8 rc=$?"; tail -3 $O/dist8.log
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu > $O/benchc.log 2>&1; echo "benchc rc=$?"; tail -3 $O/benchc.log
timeout 3000 python -m pytest tests -q -m gpu --deselect tests/test_gpu_spmd.py --deselect tests/test_gpu_bench_contract.py > $O/all.log 2>&1; echo "all rc=$?"; tail -5 $O/all.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -- python $R/bench.py --no-cpu --no-beyond-cache > $R/$O/kstats.log 2>&1
cd $R
cp $(ls $O/kstats/*/*kernel_stats.csv | head -1) $O/r04a_kernel_stats.csv 2>/dev/null
head -6 $O/r04a_kernel_stats.csv
find $O -name "*.csv" -size +3M -delete
du -sh $O
