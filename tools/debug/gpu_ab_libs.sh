#!/bin/bash
# A/B of two (or more) diagnostic builds on ONE box, alternating: tools/debug/gpu_ab_libs.sh libA.so libB.so [steps]
S=${3:-20}
for rep in 1 2; do
  for lib in "$1" "$2"; do
    echo "== $lib"
    ALTRO_HIP_LIB=$lib python tools/gpu_makespan.py $S | grep -E "wave cycles|mean wave|kernel ms|each"
  done
done
