#!/bin/bash
# k-steps per LDS stage of the 4-tile compact cross-product blocks (PLSX_CKT, default 3): builds libplsx variants with 2
# and 6 into tools/bin/ckt/ (cross-compiled here; they travel with the snapshot).  On the GPU box:
#   for v in base kt2 kt6; do python tools/bench_lib.py $v --config c4 --no-configs --no-primal --steps 5 --warmup 2; done
set -e
cd "$(dirname "$0")/.."
OUT=tools/bin/ckt; mkdir -p $OUT
OBJ=pypyls_amd/csrc/build
OTHERS=$(ls $OBJ/*.o | grep -v "plsx_xprod.o\|plsx_split.o\|plsx_compact.o")
build() {
    name=$1; kt=$2
    for u in plsx_xprod plsx_split plsx_compact; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DPLSX_CKT=$kt -c pypyls_amd/csrc/$u.hip -o $OUT/${name}_$u.o &
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $OTHERS $OUT/${name}_plsx_xprod.o $OUT/${name}_plsx_split.o $OUT/${name}_plsx_compact.o -ldl -o $OUT/libplsx_$name.so
    rm $OUT/${name}_*.o
}
build kt2 2
build kt6 6
ls -la $OUT
