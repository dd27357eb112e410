#!/bin/bash
# Build a variant of libacx_hip.so with extra compiler flags into variants/libacx_<name>.so (git-ignored; travels to the
# GPU box with the snapshot): same-box A/B runs through ACX_LIB (tools/gpu_ab.sh).
# usage: tools/build_variant.sh <name> [-DFLAG=V ...]
set -eu
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p variants
C=ahocorasick_rs_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude "$@" -o variants/libacx_$name.so \
  $C/kernels.hip $C/acx_api.cpp $C/automaton.cpp $C/comm.cpp -ldl
echo "built variants/libacx_$name.so ($*)"
