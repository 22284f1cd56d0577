"""MI355X-native batched ALTRO (AL-iLQR) MPC solver -- host-side package.

The compute path is the HIP shared library built from csrc/ and reached through the C-ABI
declared in include/altro_batch.h.  This package mirrors the Altro.jl /
TrajectoryOptimization.jl call surface the reference's benchmark scripts use.
"""
from . import problems  # noqa: F401
