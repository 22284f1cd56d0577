#!/bin/bash
# Diagnostic build of the library with the per-phase cycle stamps (-DALTRO_PHASE_STAMPS): a second .so next to the
# shipped one; use it with ALTRO_HIP_LIB=altro-mpc-icra2021_amd/csrc/libaltro_hip_stamps.so tools/gpu_makespan.py
set -e
cd "$(dirname "$0")/../altro-mpc-icra2021_amd/csrc"
python3 gen_dpp_blocks.py dpp_blocks.inc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 \
  -DALTRO_PHASE_STAMPS -DALTRO_WIDE_SINGLE_TU "$@" -o libaltro_hip_stamps.so altro_batch.hip
