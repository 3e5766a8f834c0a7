#!/bin/bash
# evidence for DESIGN.md section 4 "a whole step in one launch": step kernel vs three stage launches (eager swe2d_advance),
# with source terms, and per rank inside replayed graphs (tools/rankbench.py); kernel-trace stats of a small-mesh run
set -u
O=gpurun_out/evidence_step; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
echo "# tools/stepbench.py: us/step, three stage launches vs one step launch (tile cells, workgroup lanes), same process and box"
timeout 600 python tools/stepbench.py --sizes 50x25,100x50,177x88,250x125,300x150,354x177,500x250,1000x500 --steps 400 --tiles 128,256 160,256 256,384 2>/dev/null | grep '^{'
echo "# ... with Manning drag + Coriolis + wind stress (SRC variants)"
timeout 300 python tools/stepbench.py --sources --sizes 100x50,250x125,354x177 --tiles 128,256 2>/dev/null | grep '^{'
echo "# tools/rankbench.py: one rank, p2p halo (loopback), graph-replayed cycles, m = 4: stage launches (fused false) vs step launches"
for a in "--world 16 --rank 7 --every 4 --exchange p2p --nosplit --fused 0" "--world 16 --rank 7 --every 4 --exchange p2p --nosplit --fused 1" \
         "--world 8 --rank 3 --every 4 --exchange p2p --nosplit --fused 0" "--world 8 --rank 3 --every 4 --exchange p2p --nosplit --fused 1" \
         "--world 8 --rank 3 --nx 500 --every 4 --exchange p2p --nosplit --fused 0" "--world 8 --rank 3 --nx 500 --every 4 --exchange p2p --nosplit --fused 1"; do
  timeout 200 python tools/rankbench.py $a 2>/dev/null | tail -1
done
} > $O/r02g_step_kernel.txt
cat $O/r02g_step_kernel.txt | cut -c1-260
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -- python $R/tools/stepbench.py --sizes 100x50,250x125 --steps 400 --tiles 128,256 > $R/$O/kstats.log 2>&1
cd $R
cp $(ls $O/kstats/*/*kernel_stats.csv | head -1) $O/r02g_kernel_stats_step_small_meshes.csv 2>/dev/null; head -4 $O/r02g_kernel_stats_step_small_meshes.csv
find $O -name "*.csv" -size +3M -delete
