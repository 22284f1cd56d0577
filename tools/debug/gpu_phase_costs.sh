#!/bin/bash
# Per-call phase costs (diagnostic build) at several batch sizes: tools/debug/gpu_phase_costs.sh lib_stamps.so
export ALTRO_HIP_LIB=${1:-altro-mpc-icra2021_amd/csrc/libaltro_hip_stamps.so}
for b in 1024 8192; do
  echo "=== batch $b"
  python3 tools/gpu_makespan.py 20 $b | grep -E "wave cycles|each|kernel ms|^wave"
done
