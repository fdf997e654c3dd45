#!/bin/bash
# Builds an A/B variant of the whole library (all three kernel translation units with extra -D flags) into
# build/variants/NAME.so; load it with PT_AMD_LIB=build/variants/NAME.so (capi.py).  The in-tree library is untouched.
# usage: tools/build_variant.sh NAME "-DPT_BOUNCE_WAVES=4 ..." [GIT_REV]
# With GIT_REV the kernel sources (pt_kernels.hip, its *.inc files, headers) are taken from that commit instead of the
# working tree (old-vs-new A/B in one gpurun call); the host objects are the current ones, so the C ABI must match.
set -e
NAME=$1; FLAGS=$2; REV=${3:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/cosc_4397_pathtracing_raytracing_project_amd/csrc
KSRC=$SRC
if [ -n "$REV" ]; then
  KSRC=$ROOT/build/variants/src_$NAME
  mkdir -p $KSRC
  rm -f $KSRC/*
  for f in $(git -C $ROOT ls-tree --name-only $REV cosc_4397_pathtracing_raytracing_project_amd/csrc/ | grep -E '\.(hip|inc|h)$'); do  # the revision's own kernel sources
    git -C $ROOT show $REV:$f > $KSRC/$(basename $f)
  done
fi
OUT=$ROOT/build/variants
mkdir -p $OUT/obj_$NAME
make -C $SRC -j8 all >/dev/null
K="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize $FLAGS"
$K -ffp-contract=off -DPT_ARITH=0 -c $KSRC/pt_kernels.hip -o $OUT/obj_$NAME/k0.o &
$K -ffp-contract=fast-honor-pragmas -DPT_ARITH=1 -c $KSRC/pt_kernels.hip -o $OUT/obj_$NAME/k1.o &
$K -ffp-contract=fast-honor-pragmas -DPT_ARITH=2 -c $KSRC/pt_kernels.hip -o $OUT/obj_$NAME/k2.o &
API=$SRC/build/pt_api.o
if echo "$FLAGS" | grep -q PT_ABLATE; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPT_ABLATE -x hip -c $SRC/pt_api.cpp -o $OUT/obj_$NAME/api.o
  API=$OUT/obj_$NAME/api.o
fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/$NAME.so $OUT/obj_$NAME/k0.o $OUT/obj_$NAME/k1.o $OUT/obj_$NAME/k2.o $API \
  $SRC/build/pt_group.o $SRC/build/pt_scene.o $SRC/build/pt_image.o $SRC/build/pathtrace_shim.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
rm -rf $OUT/obj_$NAME
echo "$OUT/$NAME.so"
