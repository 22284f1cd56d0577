#!/bin/bash
# Phase breakdown of the slowest waves of the driver's 20-step launch, and the per-turn trace of the slowest one
# (needs tools/build_stamps.sh).  Output: gpurun_out/tail_profile.txt; small batches (waves alone on their SIMD / CU,
# idle memory system) beside it: gpurun_out/tail_profile_small.txt
set -e
export ALTRO_HIP_LIB=altro-mpc-icra2021_amd/csrc/libaltro_hip_stamps.so
O=gpurun_out/tail_profile.txt
python3 tools/gpu_makespan.py 20 > $O 2>&1
W=$(grep -m1 '^wave *[0-9]' $O | awk '{print $2}' | tr -d ':')
echo "slowest wave: $W" >> $O
ALTRO_DEBUG_TRACE_WAVE=$W python3 tools/debug/gpu_turn_trace.py >> $O 2>&1
for b in 64 1024 4096; do
  echo "=== batch $b" >> gpurun_out/tail_profile_small.txt
  python3 tools/gpu_makespan.py 20 $b >> gpurun_out/tail_profile_small.txt 2>&1
done
