#!/bin/bash
# Builds the WHOLE library of the working tree with extra make variables (e.g. REC=-DPT_REC_TAGGED=1, which changes host
# and kernels alike) into build/variants/NAME.so.   usage: tools/build_flags.sh NAME "REC=-DPT_REC_TAGGED=1"
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$ROOT/build/variants/flags_$NAME
rm -rf $W && mkdir -p $W/cosc/csrc $W/include
cp $ROOT/cosc_4397_pathtracing_raytracing_project_amd/csrc/*.{cpp,h,hip,inc} $ROOT/cosc_4397_pathtracing_raytracing_project_amd/csrc/Makefile $W/cosc/csrc/
cp $ROOT/include/* $W/include/
make -C $W/cosc/csrc -j8 "$@" ../libpt_amd.so >/dev/null
cp $W/cosc/libpt_amd.so $ROOT/build/variants/$NAME.so
rm -rf $W
echo "$ROOT/build/variants/$NAME.so"
