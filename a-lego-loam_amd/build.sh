#!/bin/bash
# Builds libalego_mi355x.so (HIP, gfx950) and libalego_synth.so (host) in-tree.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value"
# development builds: ALEGO_EXTRA_FLAGS=-DALEGO_TIMING ALEGO_BUILD_DIR=build_t ALEGO_SO=libalego_timing.so (then ALEGO_LIB=.../libalego_timing.so for tools/*_timing.py)
FLAGS="$FLAGS ${ALEGO_EXTRA_FLAGS:-}"
BD=${ALEGO_BUILD_DIR:-build}
SO=${ALEGO_SO:-libalego_mi355x.so}
mkdir -p $BD
pids=()
for f in kernels_ip kernels_ipf kernels_ipb kernels_fe kernels_fe2 kernels_lo kernels_lm kernels_map kernels_icp kernels_voxel lm_host alego_api; do
  if [ ! -f $BD/$f.o ] || [ csrc/$f.hip -nt $BD/$f.o ] || [ -n "$(find csrc ../include -name '*.h' -newer $BD/$f.o)" ]; then
    $HIPCC $FLAGS -c csrc/$f.hip -o $BD/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
g++ -O2 -std=c++17 -fPIC -Wall -c csrc/pc2.cpp -o $BD/pc2.o
g++ -O2 -std=c++17 -fPIC -Wall -c csrc/rosbag.cpp -o $BD/rosbag.o
$HIPCC $FLAGS -c csrc/guard_alloc.cpp -o $BD/guard_alloc.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $SO $BD/kernels_ip.o $BD/kernels_ipf.o $BD/kernels_ipb.o $BD/kernels_fe.o $BD/kernels_fe2.o $BD/kernels_lo.o $BD/kernels_lm.o $BD/kernels_map.o $BD/kernels_icp.o $BD/kernels_voxel.o $BD/lm_host.o $BD/alego_api.o $BD/pc2.o $BD/rosbag.o $BD/guard_alloc.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
g++ -O2 -std=c++17 -fPIC -shared -o libalego_synth.so csrc/synth.cpp
echo "built: $(ls -la $SO libalego_synth.so | awk '{print $9, $5}' | tr '\n' ' ')"
