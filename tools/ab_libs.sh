#!/bin/bash
# A/B on the GPU box: bench.py with each given library (PT_AMD_LIB), alternating, ROUNDS times.
# usage: tools/ab_libs.sh "bench args" ROUNDS lib1.so lib2.so ...   (use "-" for the in-tree library)
ARGS=$1; ROUNDS=$2; shift; shift
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset PT_AMD_LIB; else export PT_AMD_LIB=$(readlink -f $lib); fi
    python3 bench.py --no-extras $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$lib', d['config']['arith'], d['value'], 'k_bounce_us', d['roofline']['avg_launch_us'], 'grid', d['config']['grid_blocks'])"
  done
done
