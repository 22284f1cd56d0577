#!/bin/bash
# A/B of environment switches on one library: tools/debug/gpu_env_windows.sh lib.so "tag:VAR=val VAR2=val" ...
LIB=$1; shift
for rep in 1 2; do
  for spec in "$@"; do
    tag=${spec%%:*}; envs=${spec#*:}
    env ALTRO_HIP_LIB=$LIB $envs python3 tools/debug/gpu_lib_windows.py $tag
  done
done
