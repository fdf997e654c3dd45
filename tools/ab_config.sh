#!/bin/bash
# A/B several builds of libpt_amd.so on ONE box with tools/run_config.py (any scene), interleaved.
# usage: tools/ab_config.sh "run_config args" build/libA.so build/libB.so ...
ARGS=$1; shift
DST=cosc_4397_pathtracing_raytracing_project_amd/libpt_amd.so
cp $DST /tmp/orig.so
for round in 1 2; do
  for lib in "$@"; do
    cp $lib $DST
    echo "round $round $(basename $lib): $(python tools/run_config.py $ARGS 2>&1 | grep -E 'Msamples|bit-exact' | sed -E 's/; live rays.*dominant kernel/; dominant kernel/; s/; K=.*//' | tr '\n' ' ')"
  done
done
cp /tmp/orig.so $DST
