#!/bin/bash
# round 6, second run: the tests that are new or changed since the first (fused pair with sources / coupled / partitions, all three
# stages in one launch, stale stage buffers, options, the broken-transport path, first_contact), then what the fused paths are worth:
# kbench at 250 k ... 4 M cells with SWE2D_OPT_FUSED_STAGES = 0 / default / 3, the cfg rows with and without fusion, ranks of 2 / 4
set -u
TAG=r06b
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest -q -m gpu tests/test_gpu_parity.py -k "fused or stage_solutions or options_are" > $O/tests_fused.log 2>&1; echo "fused tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_fused.log | tail -20 | cut -c1-250
timeout 1800 python -m pytest -q -m gpu tests/test_distributed.py -k "fused_stage_pair or peer_to_peer_halos" tests/test_gpu_spmd.py::test_a_transport_that_cannot_be_set_up_is_left_by_all_ranks_together tests/test_gpu_bench_contract.py::test_first_contact_tells_the_story_of_a_multi_rank_run tests/test_gpu_bench_contract.py::test_bench_survives_a_transport_that_fails tests/test_gpu_flow_kernel.py tests/test_gpu_tracer.py > $O/tests_dist.log 2>&1; echo "dist tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_dist.log | tail -20 | cut -c1-250
kb() { timeout 300 python tools/kbench.py "$@" 2>&1 | tail -1 >> $O/${TAG}_fused_sizes.txt; }
for sz in "500 250" "707 354" "1000 500" "1414 707" "2000 1000"; do
  set -- $sz
  for f in 0 1 3; do
    THETIS_AMD_FUSE12=$f kb --nx $1 --ny $2 --steps 40 --prewarm 0.5 --tag fuse$f
  done
done
sed 's/"order.*"n_cells"/"n_cells"/; s/"vol".*//' $O/${TAG}_fused_sizes.txt | cut -c1-260
timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" > $O/${TAG}_cfgs.txt
CFGBENCH_ONLY=tracers THETIS_AMD_FUSE12=0 timeout 600 python tools/cfgbench.py 2>&1 | grep "^{" > $O/${TAG}_cfgs_nofuse.txt
sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/${TAG}_cfgs.txt | cut -c1-200; echo "--- without fusion"; sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/${TAG}_cfgs_nofuse.txt | cut -c1-200
rb() { timeout 400 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/${TAG}_rank.txt; }
for f in 0 1; do
  export THETIS_AMD_FUSE12=$f
  [ $f = 1 ] && unset THETIS_AMD_FUSE12
  rb --case cfg2 --world 4 --rank 1 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
  rb --case cfg2 --world 2 --rank 0 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
  rb --case cfg2 --world 4 --rank 1 --every 2 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
  rb --case cfg4 --world 2 --rank 0 --every 2 --exchange p2p --graph-mode full --steps 480
done
unset THETIS_AMD_FUSE12
rb --case cfg2 --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
rb --case cfg2 --world 8 --rank 3 --every 4 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
sed 's/"exchange.*"rank"/ "rank"/' $O/${TAG}_rank.txt | cut -c1-260
du -sh $O
