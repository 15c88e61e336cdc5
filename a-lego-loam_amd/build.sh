#!/bin/bash
# Builds libalego_mi355x.so (HIP, gfx950) and libalego_synth.so (host) in-tree.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value"
mkdir -p build
pids=()
for f in kernels_ip kernels_fe kernels_lo kernels_lm kernels_map kernels_icp kernels_voxel lm_host alego_api; do
  if [ ! -f build/$f.o ] || [ csrc/$f.hip -nt build/$f.o ] || [ -n "$(find csrc ../include -name '*.h' -newer build/$f.o)" ]; then
    $HIPCC $FLAGS -c csrc/$f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
g++ -O2 -std=c++17 -fPIC -Wall -c csrc/pc2.cpp -o build/pc2.o
g++ -O2 -std=c++17 -fPIC -Wall -c csrc/rosbag.cpp -o build/rosbag.o
$HIPCC $FLAGS -c csrc/guard_alloc.cpp -o build/guard_alloc.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libalego_mi355x.so build/kernels_ip.o build/kernels_fe.o build/kernels_lo.o build/kernels_lm.o build/kernels_map.o build/kernels_icp.o build/kernels_voxel.o build/lm_host.o build/alego_api.o build/pc2.o build/rosbag.o build/guard_alloc.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
g++ -O2 -std=c++17 -fPIC -shared -o libalego_synth.so csrc/synth.cpp
echo "built: $(ls -la libalego_mi355x.so libalego_synth.so | awk '{print $9, $5}' | tr '\n' ' ')"
