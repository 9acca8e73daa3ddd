#!/bin/bash
# The profiling builds of the library that tools/gpu_clocks.py (RP_CLOCKS=1: k_solve2 waves) and tools/gpu_clocks_prep.py (RP_CLOCKS=2: k_prep2 phases) load through
# RP_PLAYROOM_LIB.  Built here (hipcc cross-compiles), they travel to the GPU box with the snapshot.
set -e
cd "$(dirname "$0")/../roboticsplayroompybullet_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS -DRP_BUILD_ID=\"clocks1\" -DRP_CLOCKS=1 -shared -o ../../tools/clocks1.so rp_playroom.hip 2>/dev/null &
/opt/rocm/bin/hipcc $FLAGS -DRP_BUILD_ID=\"clocks2\" -DRP_CLOCKS=2 -shared -o ../../tools/clocks2.so rp_playroom.hip 2>/dev/null &
wait
ls -la ../../tools/clocks1.so ../../tools/clocks2.so
