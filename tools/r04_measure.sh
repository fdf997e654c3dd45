#!/bin/bash
# The round's measurement set in one gpurun call (everything lands under gpurun_out/$1/; copy what is to be judged into
# profiles/ afterwards with tools/r04_collect.sh).  usage: tools/r04_measure.sh OUTNAME [quick]
set -u
O=gpurun_out/$1; QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p $O
export TMPDIR=/tmp
# 1. PMC passes per arithmetic mode -> dominant_kernel_traffic.json (bench.py reads it): k_paths per path / per 64 rays, the
#    whole batch per sample
for M in fast exact fma; do
  (cd /tmp && timeout -k 10 500 bash $R/tools/pmc_passes.sh $O/pmc_$M --steps 250 --warmup 25 --arith $M > $R/$O/pmc_$M.log 2>&1)
  python3 tools/pmc_traffic.py $O/pmc_$M k_paths $M > $O/traffic_$M.json 2> $O/traffic_$M.err
  python3 tools/pmc_summary.py $O/pmc_$M > $O/pmc_summary_$M.txt 2>&1
  rm -f $O/pmc_$M/pass*/*.db
done
cp profiles/dominant_kernel_traffic.json $O/dominant_kernel_traffic.json
# 2. kernel stats of a bench run (rocprofv3 --kernel-trace --stats) + the bench line of that same run
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -o t -- python3 $R/bench.py --steps 1200 --warmup 25 > $R/$O/kernel_stats_bench_line.json 2> $R/$O/kstats.err)
rm -f $O/kstats/*.db
# 3. the driver's command line, and the default run (5000 spp, all legs) with the C5 line
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style_20spp.json 2> $O/bench20.err
if [ -z "$QUICK" ]; then
  timeout -k 10 600 python3 bench.py --stress > $O/bench_default_fast_5000spp_stress.json 2> $O/bench_default.err
fi
# 4. the scene-size ladder
timeout -k 10 400 python3 tools/scene_ladder.py --spp 100 > $O/scene_ladder.log 2>&1
# 5. C5 (10,170 primitives): PMC passes, grid forced so that the init-time probe adds no launches
(cd /tmp && timeout -k 10 500 bash $R/tools/pmc_config.sh $O/c5_pmc stress --spp 100 --arith fast --debug-flags 256 > $R/$O/c5_pmc.log 2>&1)
python3 tools/pmc_summary.py $O/c5_pmc > $O/c5_grid_pmc_summary_fast.txt 2>&1
rm -f $O/c5_pmc/pass*/*.db $O/c5_pmc/trace/*.db
ls $O | head -40
