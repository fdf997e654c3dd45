#!/bin/bash
# Parity fuzz in one gpurun call: default structure choice, BVH scan forced (debug_flags 512), grid forced (256, + mesh scenes), large scenes.
# usage: tools/r04_fuzz.sh OUTNAME FIRST_SEED [COUNT=1500]
O=gpurun_out/$1; S=$2; N=${3:-1500}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p $O
timeout -k 10 500 python3 tools/fuzz_parity.py $S $N > $O/f_default.log 2>&1 &
timeout -k 10 500 python3 tools/fuzz_parity.py $((S + 10000)) $((N * 2 / 3)) 512 > $O/f_scan.log 2>&1 &
timeout -k 10 500 python3 tools/fuzz_parity.py $((S + 20000)) $((N / 3)) 256 > $O/f_grid.log 2>&1 &
timeout -k 10 500 python3 tools/fuzz_parity.py $((S + 30000)) $((N / 25)) 0 large > $O/f_large.log 2>&1 &
wait
for f in $O/f_*.log; do echo "$f: $(tail -1 $f)"; done
