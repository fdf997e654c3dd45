#!/bin/bash
# Iterations per batch x k_primary pieces at N = 1 (bench.py --no-extras, 2000 steps after 200): Msamples/s.  usage: tools/sweep_k.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
for K in 20 25 40 60; do
  for P in auto 2 4; do
    if [ $P = auto ]; then unset PT_PRIMARY_PIECES; else export PT_PRIMARY_PIECES=$P; fi
    v=$(timeout -k 10 100 python3 $R/bench.py --no-extras --no-kernel-events --steps 2000 --warmup 200 --iters-per-batch $K 2>/dev/null | python3 -c 'import sys,json; print(json.loads(sys.stdin.readline())["value"])')
    echo "K=$K pieces=$P: $v"
  done
done
