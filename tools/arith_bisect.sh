#!/bin/bash
# Builds variants of the library whose FAST kernels have exactly one component of the fast mode switched on (and one
# with all of them on / all off) into build/variants/, for tools/arith_flips.py to run on the GPU box:
#   tools/arith_bisect.sh && gpurun -- 'python3 tools/arith_flips.py build/variants/*.so'
# The in-tree product library is never touched (capi honours PT_AMD_LIB).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/cosc_4397_pathtracing_raytracing_project_amd/csrc
OUT=$ROOT/build/variants
mkdir -p $OUT
make -C $SRC -j8 all >/dev/null
COMPS="TRIG DIV SQRT SLAB MV RENORM"
build() {  # name, flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -ffp-contract=fast-honor-pragmas \
    -DPT_ARITH=2 $2 -c $SRC/pt_kernels.hip -o $OUT/k_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/$1.so $SRC/build/pt_kernels_exact.o $SRC/build/pt_kernels_fma.o $OUT/k_$1.o \
    $SRC/build/pt_api.o $SRC/build/pt_group.o $SRC/build/pt_scene.o $SRC/build/pt_image.o $SRC/build/pathtrace_shim.o \
    -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  rm $OUT/k_$1.o
}
alloff=""
for c in $COMPS; do alloff="$alloff -DPT_FAST_$c=0"; done
build none "$alloff" &
build all "" &
wait
for c in $COMPS; do
  build only_$c "$(echo $alloff | sed "s/-DPT_FAST_$c=0//")" &
done
build qo_ref "-DPT_FAST_QO=0" &
build qo_ref_nomv "-DPT_FAST_QO=0 -DPT_FAST_MV=0" &
wait
ls -la $OUT
