#!/bin/bash
# Builds the WHOLE library of a git revision (host objects included, so C-ABI / layout changes between the revision and
# the working tree do not matter) into build/variants/NAME.so for old-vs-new A/B runs (tools/ab_libs.sh, PT_AMD_LIB).
# usage: tools/build_rev.sh NAME GIT_REV
set -e
NAME=$1; REV=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$ROOT/build/variants/rev_$NAME
rm -rf $W && mkdir -p $W
git -C $ROOT archive $REV cosc_4397_pathtracing_raytracing_project_amd/csrc include | tar -x -C $W
make -C $W/cosc_4397_pathtracing_raytracing_project_amd/csrc -j8 ../libpt_amd.so >/dev/null
cp $W/cosc_4397_pathtracing_raytracing_project_amd/libpt_amd.so $ROOT/build/variants/$NAME.so
rm -rf $W
echo "$ROOT/build/variants/$NAME.so"
