#!/bin/bash
# Root-causing wave_jacobi_cols<16>: builds libplsx variants whose k_sd_step<1, true, 16> runs the
# 16-rows-per-lane Jacobi under different code-generation conditions into tools/bin/jac16/ (cross-compiled
# here, they travel with the snapshot), to be run on the GPU box by tools/jacobi16_probe.py.
#   v0 the variant as it was   v1 __shfl_xor instead of DPP butterflies   v3 explicit s_waitcnt lgkmcnt(0) before wave_sync
#   v8 amdgpu_waves_per_eu(1, 2) on the Jacobi kernels (the attribute of the build in which the variant was first seen wrong)
#   v4 -O1   v5 20 rows per lane   v7 36 rows per lane inside the SAME <1, true, 16> kernel   base: the shipped library
set -e
cd "$(dirname "$0")/.."
OUT=tools/bin/jac16; mkdir -p $OUT
OBJ=pypyls_amd/csrc/build
OTHERS=$(ls $OBJ/*.o | grep -v plsx_simpls_api)
build() {  # name, flags...
    name=$1; shift
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-function "$@" -c pypyls_amd/csrc/plsx_simpls_api.hip -o $OUT/$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $OTHERS $OUT/$name.o -ldl -o $OUT/libplsx_$name.so
    rm $OUT/$name.o
}
build v0 -O3 &
build v1 -O3 -DPLSX_JV=1 &
build v3 -O3 -DPLSX_JV=3 &
build v4 -O1 &
build v5 -O3 -DPLSX_JV=5 &
build v7 -O3 -DPLSX_JV=7 &
build v8 -O3 -DPLSX_JV=8 &
wait
ls -la $OUT
