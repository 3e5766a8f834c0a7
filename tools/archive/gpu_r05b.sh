#!/bin/bash
# round 5, second run: the flow kernel after the chain cuts (batched LDS reads in publish / poll / receive / push, cell integrals
# before the polling loop), the granules' check word, adversary builds (delay, tear, tear without check = negative control),
# periodic verification, soak; ranks of eight with per-block time stamps
set -u
O=gpurun_out/r05b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_flow_kernel.py tests/test_distributed.py tests/test_gpu_fuzz.py -q -m gpu -x > $O/flow_tests.log 2>&1; echo "flow tests rc=$?"; tail -3 $O/flow_tests.log
timeout 1200 python -m pytest tests/test_gpu_spmd.py tests/test_gpu_bench_contract.py -q -m gpu -x -k "not full_size" > $O/spmd_tests.log 2>&1; echo "spmd + bench contract rc=$?"; tail -3 $O/spmd_tests.log
THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_delay.so timeout 900 python -m pytest tests/test_gpu_flow_kernel.py tests/test_distributed.py -m gpu -q -k "lags or lagging" 2>&1 | tail -2 | sed "s/^/[delay] /"
for v in tear tear_nocheck; do
  THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_$v.so timeout 900 python -m pytest tests/test_gpu_flow_kernel.py tests/test_distributed.py tests/test_gpu_spmd.py -m gpu -q \
      -k "two_halves or torn or periodic_verification" 2>&1 | tail -4 | sed "s/^/[$v] /"
done
rb() { timeout 300 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/rank.txt; }
rb --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --world 8 --rank 0 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --world 8 --rank 3 --every 1 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --world 8 --rank 3 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 1920
rb --world 16 --rank 7 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
THETIS_AMD_LIB=$PWD/build_dbg/libswe2d_latecell.so timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 2>&1 | tail -1 | sed 's/^/LATE CELL TERMS: /' >> $O/rank.txt
cut -c1-20,230- $O/rank.txt
for S in 6 8; do
  THETIS_AMD_LIB=$PWD/build_dbg/wt$S.so RANKBENCH_TIMING_DUMP=$O/stamps_$S.npz timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920 --timing 2>&1 | tail -2 | sed "s/^/stage $S: /" >> $O/rank_timing.txt
done
cut -c1-1200 $O/rank_timing.txt
for nx in 125 354; do
  THETIS_AMD_LIB=$PWD/build_dbg/wt8.so timeout 300 python tools/flowtiming.py --nx $nx --ny $((nx/2)) > $O/flow_timing_$nx.json 2> $O/t.err
done
for fl in 0 1; do THETIS_AMD_FLOW=$fl timeout 300 python tools/kbench.py --nx 354 --ny 177 --steps 384 --prewarm 0.5 --tag flow$fl 2>&1 | tail -1 >> $O/flow_sizes.txt; done
cat $O/flow_sizes.txt | cut -c1-300
du -sh $O
