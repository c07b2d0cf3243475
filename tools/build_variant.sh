#!/bin/bash
# Development: build a variant of libalproj_hip.so with extra -D flags for alp_raster.hip (always with -DALP_DEV:
# the library then reports its switches through alp_build_flags() and tests / bench.py refuse it).
#   tools/build_variant.sh NAME -DFOO=1 ...   ->  build/abl/libalproj_NAME.so   (use with ALPROJ_HIP_LIB)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Iinclude -Ialproj_amd/csrc \
    -ffp-contract=off -DALP_DEV "$@" -c alproj_amd/csrc/alp_raster.hip -o build/abl/raster_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libalproj_$name.so build/alp_core.o build/alp_points.o \
    build/abl/raster_$name.o build/alp_mesh.o build/alp_rasterize.o build/alp_sampler.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo build/abl/libalproj_$name.so
