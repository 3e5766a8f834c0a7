#!/bin/bash
# The gpurun session of round 6.  Usage (on the GPU box, from the repo root):
#   TAG=r06a STEPS="tests bench kstats pmc1m pmc4m cfgs ranks" bash tools/gpu_session.sh
# leaves its files under gpurun_out/$TAG (copy what is to be judged into profiles/).  Steps:
#   tests   the whole GPU suite (+ smoke)
#   bench   the bench line as the driver runs it
#   kstats  rocprofv3 --kernel-trace --stats of `bench.py --no-cpu --no-beyond-cache`: every row is the 1 M-cell workload
#   pmc1m   PMC passes of the library's launches at 1 M cells -> ${TAG}_traffic.json (+ VALU wave-instructions per step)
#   pmc4m   the same at 4 M cells -> ${TAG}_traffic_4m.json
#   cfgs    every row of tools/cfgbench.py
#   ranks   ranks of 8 / 4 / 2 on one GPU (tools/rankbench.py)
set -u
TAG=${TAG:-r06a}
STEPS=${STEPS:-"tests bench kstats pmc1m pmc4m cfgs ranks"}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
has() { case " $STEPS " in *" $1 "*) return 0;; *) return 1;; esac; }

if has tests; then
  timeout 3000 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | tail -40 | cut -c1-220
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
fi
if has bench; then
  timeout 900 python bench.py > $O/${TAG}_bench_line.json 2> $O/bench.err; tail -1 $O/${TAG}_bench_line.json | cut -c1-1500
fi
if has kstats; then
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -- python $R/bench.py --no-cpu --no-beyond-cache > $R/$O/kstats.log 2>&1
  cd $R
  cp $(ls $O/kstats/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_1m.csv 2>/dev/null
  head -6 $O/${TAG}_kernel_stats_1m.csv | cut -c1-220
  find $O/kstats -name "*kernel_trace.csv" -delete
fi
if has pmc1m; then
  bash tools/pmc.sh $R/$O/pmc1m python $R/tools/kbench.py --steps 4 --order auto --calibrate > /dev/null 2>&1
  cd $R
  python tools/pmc_summary.py $O/pmc1m swe_ > $O/${TAG}_pmc_summary.txt 2>&1
  python tools/make_traffic_json.py $O/pmc1m 1000000 $O/${TAG}_traffic.json "bench workload (1M triangles), round-6 library ($TAG)" > /dev/null 2>&1
  grep -E "traffic_bytes|algorithmic_bytes_per|valu_wave" $O/${TAG}_traffic.json
  rm -rf $O/pmc1m
fi
if has pmc4m; then
  bash tools/pmc.sh $R/$O/pmc4m python $R/tools/kbench.py --nx 2000 --ny 1000 --steps 3 --prewarm 0.2 --order auto --calibrate > /dev/null 2>&1
  cd $R
  python tools/pmc_summary.py $O/pmc4m swe_ > $O/${TAG}_pmc_summary_4m.txt 2>&1
  python tools/make_traffic_json.py $O/pmc4m 4000000 $O/${TAG}_traffic_4m.json "RectangleMesh(2000,1000) = 4M triangles (roofline.beyond_cache), round-6 library ($TAG)" > /dev/null 2>&1
  grep -E "traffic_bytes|algorithmic_bytes_per|valu_wave" $O/${TAG}_traffic_4m.json
  rm -rf $O/pmc4m
fi
if has cfgs; then
  timeout 900 python tools/cfgbench.py 2>&1 | grep "^{" > $O/${TAG}_cfgs.txt; sed 's/"algorithmic_bytes.*frac_of_8TBs/"frac/' $O/${TAG}_cfgs.txt | cut -c1-170
fi
if has ranks; then
  rb() { timeout 400 python tools/rankbench.py "$@" 2>&1 | tail -1 >> $O/${TAG}_rank.txt; }
  rb --case cfg2 --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
  rb --case cfg2 --world 8 --rank 0 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 1920
  rb --case cfg2 --world 4 --rank 1 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
  rb --case cfg2 --world 2 --rank 0 --every 4 --exchange p2p --nosplit --flow 0 --graph-mode full --steps 960
  rb --case cfg4 --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480
  rb --case cfg4_tracer_only --world 8 --rank 3 --every 2 --exchange p2p --graph-mode full --steps 480
  rb --case cfg5 --world 8 --rank 3 --every 2 --exchange p2p --nosplit --flow 1 --flowx 1 --graph-mode full --steps 960
  sed 's/"exchange.*"rank"/ "rank"/; s/"every.*"us_per_step"/ us_per_step/' $O/${TAG}_rank.txt | cut -c1-200
fi
find $O -name "*.csv" -size +3M -delete
du -sh $O
