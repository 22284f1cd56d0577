#!/bin/bash
# Experiment build of the library for tools/debug/gpu_lib_windows.py: only the headline instantiation
# (-DALTRO_DEV_HEADLINE_ONLY, well under a minute), written to tools/ab/lib_<tag>.so (git-ignored; travels to the GPU box).
#   tools/build_ab.sh tag [extra -D flags]
set -e
TAG=$1; shift
R="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$R/tools/ab"
cd "$R/altro-mpc-icra2021_amd/csrc"
python3 gen_dpp_blocks.py dpp_blocks.inc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 \
  -DALTRO_DEV_HEADLINE_ONLY "$@" -o "$R/tools/ab/lib_$TAG.so" altro_batch.hip
