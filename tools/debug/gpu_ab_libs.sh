#!/bin/bash
# A/B of diagnostic builds on ONE box, alternating, twice over: tools/debug/gpu_ab_libs.sh STEPS libA.so libB.so ...
S=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    ALTRO_HIP_LIB=$lib python tools/gpu_makespan.py $S | grep -E "wave cycles|kernel ms|each"
  done
done
