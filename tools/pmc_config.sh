#!/bin/bash
# Kernel trace + PMC passes for one tools/run_config.py configuration (any scene), one rocprofv3 pass per group.
# usage: tools/pmc_config.sh OUTDIR run_config-args...
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p "$R/$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace" -o t -- python3 "$R/tools/run_config.py" "$@" > "$R/$OUT/trace.log" 2> "$R/$OUT/trace.err" || echo "trace failed"
i=0
for grp in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_SMEM" \
  "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$R/$OUT/pass$i" -o p -- python3 "$R/tools/run_config.py" "$@" > "$R/$OUT/pass$i.log" 2> "$R/$OUT/pass$i.err" || echo "pass $i failed"
done
ls "$R/$OUT" | head -40
