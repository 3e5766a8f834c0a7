#!/bin/bash
# round 4, tenth session: the example script with the in-launch exchange of the flow kernel among 8 and 4 processes sharing the GPU
# (meshes sized so that the blocks of ALL ranks are resident together), against the single-device run
set -u
O=gpurun_out/r04j; mkdir -p $O
export TMPDIR=/tmp
export THETIS_AMD_DIST_BACKEND=gloo THETIS_AMD_DIST_TIMEOUT_S=120
for cfg in "8 320 160" "4 340 170" "8 320 160 1"; do
  set -- $cfg
  n=$1; nx=$2; ny=$3; every=${4:-}
  python examples/channel2d.py --nx $nx --ny $ny --t-end 100 2>&1 | grep -E "^volume|^ +[0-9]+ +[0-9]+ " > $O/one_$nx.txt
  ( time THETIS_AMD_SPMD_FLOW=1 THETIS_AMD_EXCHANGE_EVERY=$every timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n examples/channel2d.py --nx $nx --ny $ny --t-end 100 ) > $O/many_${n}_$nx.log 2>&1
  echo "== $n ranks, $nx x $ny, every=$every: single:"; tail -2 $O/one_$nx.txt; echo "   ranks:"; grep -E "^volume|^ +[0-9]+ +[0-9]+ |Error|error" $O/many_${n}_$nx.log | tail -4 | cut -c1-200; grep real $O/many_${n}_$nx.log
done
