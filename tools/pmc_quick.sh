#!/bin/bash
# usage: pmc2.sh OUT LIB   (LIB = - for in-tree)
OUT=$1; LIB=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
if [ "$LIB" != "-" ]; then export PT_AMD_LIB=$(readlink -f $R/$LIB); fi
mkdir -p $R/$OUT
i=0
for grp in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_SMEM" ; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$R/$OUT/pass$i" -o p -- python3 "$R/bench.py" --no-extras --steps 100 --warmup 25 ${PMC_BENCH_ARGS:-} > "$R/$OUT/pass$i.json" 2> "$R/$OUT/pass$i.err")
done
rm -f $R/$OUT/pass*/*.db
python3 $R/tools/pmc_summary.py $R/$OUT > $R/$OUT/summary.txt
