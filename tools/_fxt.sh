O=gpurun_out/r03s; mkdir -p $O; rm -f $O/t.log
for S in 6 8 10 11 12; do
  THETIS_AMD_LIB=$PWD/variants/wt$S.so timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240 --timing 2>&1 | tail -2 | sed "s/^/S=$S /" >> $O/t.log
done
THETIS_AMD_LIB=$PWD/variants/wt8.so timeout 300 python tools/rankbench.py --world 8 --rank 0 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 240 --timing 2>&1 | tail -2 | sed "s/^/rank0 S=8 /" >> $O/t.log
cat $O/t.log
