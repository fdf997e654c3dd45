#!/bin/bash
# One PMC pass (instruction counts) for a bench run with the given library; prints per-kernel means.
# usage: tools/pmc_quick.sh LIB.so OUTDIR [bench args]
LIB=$1; OUT=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cp $LIB $R/cosc_4397_pathtracing_raytracing_project_amd/libpt_amd.so
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$R/$OUT/pass1" -o p -- python3 "$R/bench.py" --no-extras --no-kernel-events "$@" > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $R/$OUT | grep -A9 "k_bounce\[all\]\|k_primary  "
