"""MI355X-native batched ALTRO (AL-iLQR) MPC solver -- host-side package.

The compute path is the HIP shared library built from csrc/ and reached through the C-ABI
declared in include/altro_batch.h.  This package mirrors the Altro.jl /
TrajectoryOptimization.jl call surface the reference's benchmark scripts use (api.py) and
restates the reference's problem generators and MPC harness (problems.py, mpc.py).
"""
from . import _lib, benchmarks, mpc, parallel, problems, results_io  # noqa: F401
from ._lib import SOLVE_SUCCEEDED, STATUS_NAMES, debug_set  # noqa: F401
from .api import alpha_trace, gains, set_dynamics, set_dynamics_track  # noqa: F401
from .api import (ALTROSolver, AltroError, BoundConstraint, ConstraintList, GoalConstraint, LinearConstraint,
                  LinearModel, NormConstraint, Problem,  # noqa: F401
                  SolverOptions, TrackingObjective, benchmark_solve, confirm_counter, reuse_counter, polish_stats, polish_dual_residuals, controls, cost, get_duals, initial_controls,
                  iterations, max_violation, set_duals, set_initial_state, set_options, set_tracking_cost, shift_fill,
                  solve, solve_counters, states, stats, status, timing_get, timing_reset, update_constraint_data, update_trajectory,
                  wave_cycles, work_counters)
