#!/bin/bash
# Copies what tools/r03_measure.sh left under gpurun_out/NAME/ into profiles/ under this round's names.
# usage: tools/r03_collect.sh NAME
set -e
S=gpurun_out/$1; P=profiles
cp $S/ubench_valu.txt $P/r03_ubench_valu.txt
cp $S/dominant_kernel_traffic.json $P/dominant_kernel_traffic.json
for M in fast exact fma; do cp $S/pmc_summary_$M.txt $P/r03_pmc_summary_$M.txt; done
cp $S/kstats/t_kernel_stats.csv $P/r03_kernel_stats.csv
cp $S/kernel_stats_bench_line.json $P/r03_kernel_stats_bench_line.json
cp $S/bench_driver_style_20spp.json $P/r03_bench_driver_style_20spp.json
[ -f $S/bench_default_fast_5000spp_stress.json ] && cp $S/bench_default_fast_5000spp_stress.json $P/r03_bench_default_fast_5000spp_stress.json
[ -f $S/c5_grid_pmc_summary_fast.txt ] && cp $S/c5_grid_pmc_summary_fast.txt $P/r03_c5_grid_pmc_summary_fast.txt
[ -f $S/c5_pmc/trace/t_kernel_stats.csv ] && cp $S/c5_pmc/trace/t_kernel_stats.csv $P/r03_c5_grid_kernel_stats_fast.csv
ls -la $P/r03_* $P/dominant_kernel_traffic.json
