#!/bin/bash
# Development: variant of libalproj_hip.so with extra -D flags for alp_points.hip (population kernels).
#   tools/build_variant_points.sh NAME -DFOO=1 ...   ->  build/abl/libalproj_NAME.so   (use with ALPROJ_HIP_LIB)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Iinclude -Ialproj_amd/csrc \
    "$@" -Rpass-analysis=kernel-resource-usage -c alproj_amd/csrc/alp_points.hip -o build/abl/points_$name.o 2> build/abl/points_$name.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libalproj_$name.so build/alp_core.o build/abl/points_$name.o \
    build/alp_raster.o build/alp_mesh.o build/alp_rasterize.o build/alp_sampler.o build/alp_host.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo build/abl/libalproj_$name.so
