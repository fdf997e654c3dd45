#!/bin/bash
# A/B several builds of the library on ONE box with tools/run_config.py (any scene), interleaved.  The builds are loaded
# through PT_AMD_LIB (capi.py); the in-tree product library is never overwritten ("-" = the in-tree library).
# usage: tools/ab_config.sh "run_config args" build/variants/A.so build/variants/B.so ...
ARGS=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset PT_AMD_LIB; else export PT_AMD_LIB=$(readlink -f $lib); fi
    echo "round $round $(basename $lib): $(python3 tools/run_config.py $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], 'Msamples/s; dominant kernel', d['dominant_kernel_us'], 'us;', d.get('rows', ''))")"
  done
done
