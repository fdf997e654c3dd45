#!/bin/bash
# Two SQ counter passes (instruction mix, waiting, LDS) for one tools/run_config.py configuration; summary on stdout.
# usage: tools/pmc_quick2.sh OUTDIR run_config-args...
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p "$R/$OUT"
i=0
for grp in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_SMEM" ; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$R/$OUT/pass$i" -o p -- python3 "$R/tools/run_config.py" "$@" > "$R/$OUT/pass$i.log" 2> "$R/$OUT/pass$i.err") || echo "pass $i failed"
  rm -f "$R/$OUT"/pass$i/*.db
done
python3 "$R/tools/pmc_summary.py" "$R/$OUT"
