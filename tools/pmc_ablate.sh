#!/bin/bash
# VALU instruction budget of k_bounce by ablation (PMC SQ_INSTS_VALU): full, without primitive tests, without bounce.
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp
for f in 0 4 8 12; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d "$R/gpurun_out/abl$f/pass1" -o p -- python3 "$R/bench.py" --no-extras --no-kernel-events --steps 100 --warmup 25 --debug-flags $f > /dev/null 2>&1
  echo "== debug_flags $f"; python3 $R/tools/pmc_summary.py $R/gpurun_out/abl$f | grep -A5 "k_bounce\[d1\]\|k_bounce\[d4\]"
done
