#!/bin/bash
# VALU instruction budget of k_bounce by ablation (PMC SQ_INSTS_VALU): full, without primitive tests (4), without the
# bounce-direction sampling (8), without both.  The ablation switches produce WRONG images and exist only in a
# -DPT_ABLATE build of the library:
#   (here)      tools/build_variant.sh ablate "-DPT_ABLATE"
#   (GPU box)   tools/pmc_ablate.sh build/variants/ablate.so [arith]
# The release library rejects these debug_flags (pt_init fails).
LIB=${1:?usage: tools/pmc_ablate.sh build/variants/ablate.so [exact|fma|fast]}; MODE=${2:-fast}
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp
export PT_AMD_LIB=$(readlink -f $LIB)
for f in 0 4 8 12; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d "$R/gpurun_out/abl$f/pass1" -o p -- python3 "$R/bench.py" --no-extras --steps 100 --warmup 25 --arith $MODE --debug-flags $f > /dev/null 2>&1
  echo "== debug_flags $f"; python3 $R/tools/pmc_summary.py $R/gpurun_out/abl$f | grep -A5 "k_bounce\[d1\]\|k_bounce\[d4\]"
  rm -f $R/gpurun_out/abl$f/pass1/*.db
done
