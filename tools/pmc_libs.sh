#!/bin/bash
# Two PMC passes (instruction mix, TA/TCP activity) of one run_config configuration for several builds of the library,
# loaded through PT_AMD_LIB (the in-tree product library is never overwritten; "-" = the in-tree library).
# usage: tools/pmc_libs.sh OUTDIR "run_config args" build/variants/A.so build/variants/B.so ...
set -u
OUT=$1; ARGS=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset PT_AMD_LIB; n=intree; else export PT_AMD_LIB=$(readlink -f $lib); n=$(basename $lib .so); fi
  mkdir -p "$R/$OUT/$n"
  i=0
  for grp in \
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
    "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" ; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$R/$OUT/$n/pass$i" -o p -- python3 "$R/tools/run_config.py" $ARGS > "$R/$OUT/$n/pass$i.log" 2> "$R/$OUT/$n/pass$i.err" || echo "pass $i failed"
  done
  rm -f "$R/$OUT/$n"/pass*/*.db
done
