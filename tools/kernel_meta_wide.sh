#!/bin/bash
# Register / scratch use and ISA of ONE instantiation of the one-wave-per-instance kernel: tools/kernel_meta_wide.sh 12 true
MC=${1:-12}; SM=${2:-true}; shift 2 2>/dev/null
T=$(mktemp -d)
cat > $T/one.hip <<EOT
#include "$(cd "$(dirname "$0")/.." && pwd)/altro-mpc-icra2021_amd/csrc/solve_wide.h"
template __global__ void altro_wide::wide_kernel<$MC, $SM>(altro_wide::Params, int, int, int);
EOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 -DALTRO_DEV_HEADLINE_ONLY "-DALTRO_DEV_WIDE_KERNEL=wide_kernel<$MC,$SM>" "$@" -o $T/one.s $T/one.hip || exit 1
grep -E "\.(vgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):" $T/one.s | tr -s ' ' | tr '\n' ' '; echo
echo "instructions: $(grep -cE '^\s+(v_|s_|ds_|global_|scratch_|buffer_)' $T/one.s)  mfma: $(grep -c v_mfma $T/one.s)  scratch ops: $(grep -c scratch_ $T/one.s) readlane: $(grep -c v_readlane $T/one.s)"
cp $T/one.s /tmp/last_wide.s
rm -rf $T
