#!/bin/bash
# compile scream_amd/csrc/proj_ring.hip to assembly (extra flags as arguments) and print the register / spill statistics of its kernels
cd "$(dirname "$0")/../scream_amd" && mkdir -p build
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function "$@" -S --cuda-device-only csrc/proj_ring.hip -o build/proj_ring.s 2>&1 | grep -v "hip-link"
grep -E "^\s+\.(vgpr_count|agpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size)|\.name:" build/proj_ring.s | grep -A5 "proj_ring"
