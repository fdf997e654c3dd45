#!/bin/bash
# A/B several builds of libpt_amd.so on ONE box, interleaved (never compare across boxes).
# usage: tools/ab.sh "bench args" build/libA.so build/libB.so ...
ARGS=$1; shift
DST=cosc_4397_pathtracing_raytracing_project_amd/libpt_amd.so
cp $DST /tmp/orig.so
for round in ${ROUNDS:-1 2 3}; do
  for lib in "$@"; do
    cp $lib $DST
    v=$(python bench.py --no-extras $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(d['value'], r.get('avg_launch_us'), r.get('frac'))")
    echo "round $round $(basename $lib): $v"
  done
done
cp /tmp/orig.so $DST
