#!/bin/bash
# Register / scratch / LDS use of the headline kernel instantiation (compiles solve_dpp16.h alone: ~6 s).
#   tools/kernel_meta.sh [NX NU CONES] [extra -D flags]
NX=${1:-12}; NU=${2:-4}; CO=${3:-false}; shift 3 2>/dev/null
python3 "$(dirname "$0")/../altro-mpc-icra2021_amd/csrc/gen_dpp_blocks.py" "$(dirname "$0")/../altro-mpc-icra2021_amd/csrc/dpp_blocks.inc"
T=$(mktemp -d)
cat > $T/one.hip <<EOT
#include "$(cd "$(dirname "$0")/.." && pwd)/altro-mpc-icra2021_amd/csrc/solve_dpp16.h"
template __global__ void altro::solve_kernel<$NX, $NU, $CO>(altro::SolveParams);
EOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Wno-unused-value "$@" -o $T/one.s $T/one.hip || exit 1
grep -E "\.(vgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):" $T/one.s | tr -s ' ' | tr '\n' ' '; echo
echo "instructions: $(grep -cE '^\s+(v_|s_|ds_|global_|scratch_|buffer_)' $T/one.s)  fmac_dpp: $(grep -c v_fmac_f64_dpp $T/one.s)  scratch ops: $(grep -c scratch_ $T/one.s)"
cp $T/one.s /tmp/last_kernel.s
rm -rf $T
