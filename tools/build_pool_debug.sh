#!/bin/bash
# builds raytracer_amd/librt_mi355x_dbg.so: the library with RT_POOL_DEBUG (bounds checks inside the pooled traversal kernel);
# use it with RT_LIB_PATH=raytracer_amd/librt_mi355x_dbg.so
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/rt_dbg
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function -DRT_POOL_DEBUG"
for f in rt_api rt_kernels rt_bvh rt_sort rt_ploc rt_sah rt_bluenoise; do
  hipcc $FLAGS -c raytracer_amd/csrc/$f.hip -o /tmp/rt_dbg/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o raytracer_amd/librt_mi355x_dbg.so /tmp/rt_dbg/*.o
ls -la raytracer_amd/librt_mi355x_dbg.so
