#!/bin/bash
# Copies the judged summaries of a tools/r04_measure.sh run from gpurun_out/$1 into profiles/ (tracked).
set -eu
O=gpurun_out/$1
cp $O/dominant_kernel_traffic.json profiles/dominant_kernel_traffic.json
cp $O/kstats/t_kernel_stats.csv profiles/r04_kernel_stats.csv
cp $O/kernel_stats_bench_line.json profiles/r04_kernel_stats_bench_line.json
for M in fast exact fma; do cp $O/pmc_summary_$M.txt profiles/r04_pmc_summary_$M.txt; done
cp $O/bench_driver_style_20spp.json profiles/r04_bench_driver_style_20spp.json
[ -f $O/bench_default_fast_5000spp_stress.json ] && cp $O/bench_default_fast_5000spp_stress.json profiles/r04_bench_default_fast_5000spp_stress.json
cp $O/scene_ladder.log profiles/r04_scene_ladder.log
cp $O/c5_grid_pmc_summary_fast.txt profiles/r04_c5_grid_pmc_summary_fast.txt
ls -la profiles | grep r04_
