#!/bin/bash
# development: gfx950 assembly of one translation unit with the flags of build.sh.  usage: tools/isa.sh kernels_lm [extra flags]  -> /tmp/isa/<name>.s
# then e.g.  awk '/^lm_knn/,/s_endpgm/' /tmp/isa/kernels_lm.s | grep -c v_pk_
cd "$(dirname "$0")/../a-lego-loam_amd"
f=$1; shift
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-value "$@" -S --cuda-device-only csrc/$f.hip -o /tmp/isa/$f.s
echo "/tmp/isa/$f.s: $(grep -c '^\s*v_' /tmp/isa/$f.s) VALU lines, $(grep -c 'v_pk_\(mul\|add\|fma\)_f32' /tmp/isa/$f.s) packed f32"
