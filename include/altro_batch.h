/*
 * altro_batch.h -- C-ABI of libaltro_hip.so: batched ALTRO (AL-iLQR) MPC solves on MI355X.
 *
 * The reference (RoboticExplorationLab/altro-mpc-icra2021) has no FFI: its benchmark scripts
 * call the exported Julia API of Altro.jl / TrajectoryOptimization.jl / RobotDynamics.jl
 * in-process.  Each entry point below replaces one of those calls for a BATCH of independent
 * MPC instances; the reference call site it mirrors is cited (paths relative to the reference
 * repo).  A Julia `ccall` shim over these symbols is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns int32 (0 = ALTRO_OK); nothing throws or aborts across the ABI;
 *     altro_last_error() returns a message for the last failing call of a handle (or of
 *     altro_batch_create when called with NULL).
 *   - solver failures are NOT errors: they are per-instance status values (enum below), as in
 *     the reference (random_linear_problem.jl:166, altro_solver.jl:81).
 *   - the caller owns every buffer it passes; the library copies in/out and keeps no caller
 *     pointer after return (Julia's GC may move or free them).
 *   - host arrays are instance-major: X is [batch][N][n], U is [batch][N-1][m]; matrices are
 *     COLUMN-major n x n / n x m blocks (Julia layout).
 *   - a handle is single-owner (not thread-safe) and owns one HIP stream and all its device
 *     memory.  Different handles may be driven from different threads.
 *   - all arithmetic is FP64, as in the reference.
 */
#ifndef ALTRO_BATCH_H
#define ALTRO_BATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  ALTRO_OK = 0,
  ALTRO_ERR_INVALID_ARG = 1,
  ALTRO_ERR_UNSUPPORTED = 2, /* problem shape/constraint outside the built kernel set */
  ALTRO_ERR_HIP = 3,         /* HIP runtime failure (message has the HIP error string) */
  ALTRO_ERR_STATE = 4,       /* call sequence error (e.g. solve before set_dynamics) */
  ALTRO_ERR_INTERNAL = 5     /* host allocation failure or any other C++ exception inside the library:
                                caught at the boundary, never propagated to the caller */
};

/* Altro.TerminationStatus.  Only UNSOLVED and SOLVE_SUCCEEDED are named in the reference
 * (simple_rocket.jl:144, random_linear_problem.jl:166); order restated from Altro.jl v0.2. */
enum {
  ALTRO_UNSOLVED = 0,
  ALTRO_SOLVE_SUCCEEDED = 1,
  ALTRO_MAX_ITERATIONS = 2,
  ALTRO_MAX_ITERATIONS_OUTER = 3,
  ALTRO_MAXIMUM_COST = 4,
  ALTRO_STATE_LIMIT = 5,
  ALTRO_CONTROL_LIMIT = 6,
  ALTRO_NO_PROGRESS = 7,
  ALTRO_COST_INCREASE = 8
};

/* constraint menu: user constraint types of the reference are Julia closures
 * (new_constraints.jl:31-62), which cannot cross a C ABI; they are all affine maps of
 * z = [x;u] into {=0, <=0, second-order cone} and are passed as data. */
enum { ALTRO_CON_BOX = 0, ALTRO_CON_LINEAR = 1, ALTRO_CON_SOC = 2 };
enum { ALTRO_SENSE_EQ = 0, ALTRO_SENSE_INEQ = 1 };

typedef struct altro_dims {
  int32_t batch; /* number of independent MPC instances */
  int32_t n;     /* state dimension   */
  int32_t m;     /* control dimension */
  int32_t N;     /* knot points       */
} altro_dims;

/* Altro.SolverOptions: fields the reference sets (run_random_linear.jl:41-49,
 * run_simple_rocket.jl:39-50, grasp_benchmark.jl:19-34, ALTROParams.jl:86-95) plus the
 * iLQR/AL internals they rely on.  altro_default_opts() fills Altro.jl's defaults. */
typedef struct altro_opts {
  double cost_tolerance;
  double cost_tolerance_intermediate;
  double gradient_tolerance;
  double gradient_tolerance_intermediate;
  double constraint_tolerance;
  double penalty_initial;  /* NaN = per-constraint default (1.0) */
  double penalty_scaling;  /* NaN = per-constraint default (10)  */
  double penalty_max;
  double dual_max;
  double line_search_lower_bound;
  double line_search_upper_bound;
  double max_cost_value;
  double max_state_value;
  double max_control_value;
  double bp_reg_initial;
  double bp_reg_increase_factor;
  double bp_reg_max;
  double bp_reg_min;
  double bp_reg_fp;
  int32_t iterations;
  int32_t iterations_inner;
  int32_t iterations_outer;
  int32_t iterations_linesearch;
  int32_t dJ_counter_limit;
  int32_t reset_duals;
  int32_t reset_penalties;
  int32_t bp_reg;
  int32_t soc_second_order;
  /* 0 (default): the 16-lane kernels take these shortcuts relative to Altro.jl's forwardpass! / backwardpass!
   *   - confirmation iterations.  The last iteration of nearly every warm MPC solve only confirms convergence: the
   *     problem is quadratic inside an active set, so at the point the previous Newton step reached the backward
   *     pass returns feedforward terms at rounding level; the reference then rolls out Z + O(|d|), finds
   *     |dJ| ~ 1e-13 and stops (step accepted, or after iterations_linesearch fruitless halvings, whichever way
   *     the rounding of J falls).  Such an iteration is booked as converged on the trajectory it holds, with the
   *     same status and iteration count, when (a) on box-constrained problems, from the second iteration of an
   *     inner solve on, the first-order costate sweep finds |l_u + B' lambda| <= 0.25e-9 dt R (1 + |u|) at every
   *     knot (so |d| <= 0.5e-9 (1 + |u|)) and the active set unchanged since the previous backward pass -- no
   *     backward pass is run, the gains reported are that pass's (the same matrices) with d = 0; or (b) a backward
   *     pass returns |d| <= 1e-9 (1 + |u|) at every knot -- its rollout, line search and gradient sweep are skipped;
   *   - a line search whose alpha = 1 trial moved no element by more than 1e-7 (1 + |z|) while the quadratic
   *     model promised less than cost_tolerance / 1000 is ended there (the reference halves alpha
   *     iterations_linesearch more times, fails the same way, and the iteration ends "converged" either way);
   *   - S is not re-symmetrised after every knot of the backward pass (the asymmetry stays at rounding level).
   * 1: none of them, the reference's exact sequence (about half the throughput on BASELINE's headline workload).
   * The one-wave-per-instance kernel (sizes outside the 16-lane set, per-knot dynamics) takes (b) and the line-search
   * early-out the same way in the default mode and always re-symmetrises S; for n, m <= 16 (at most 16 linear rows,
   * no cones) and for box-only time-invariant problems with n > 16, m <= 16 it also has a costate sweep (a), which
   * solves for the feedforward terms with the factors of Quu the last backward pass stored and applies test (b) to them.
   * For the n > 16 class it goes one step further (gain reuse): any iteration whose active set and penalty are those of
   * the stored pass takes its gains K from memory -- inside a fixed active set they do not depend on the trajectory --
   * and only its feedforward terms from that first-order pass; dynamics, cost, constraint data and options setters
   * drop the stored pass. */
  int32_t strict;
  /* Altro.SolverOptions.kickout_max_penalty (default false): with 0 the AL outer loop does not stop when the penalty
   * has reached penalty_max -- it goes on with dual updates at the cap until the constraints are satisfied or
   * iterations_outer runs out (MAX_ITERATIONS_OUTER); 1 ends the solve there (status UNSOLVED if the violation is
   * still above the tolerance). */
  int32_t kickout_max_penalty;
  /* Projected-Newton polish (Altro.jl solve!(::ALTROSolver): AL stage to projected_newton_tolerance, then
   * solve!(::ProjectedNewtonSolver) if the violation is still above constraint_tolerance; ALTRO, IROS 2019, Algorithm 4).
   * altro_default_opts() gives 0: Altro.jl's own default is true, but every script of the reference on this path sets
   * projected_newton = false (eleven occurrences), and the ccall shim passes the Julia-side value explicitly.  1: plain
   * solves (altro_batch_solve) run the polish after the AL kernel -- csrc/pn_polish.h on the 16-lane backend, csrc/pn_wide.h
   * on the one-wave-per-instance backend (any n <= 64, m <= 32, per-knot dynamics) -- primal projection first, then
   * Altro's multiplier projection (altro_batch_get_polish_dual_residuals).  Inside the device-resident MPC loop
   * (altro_mpc_step_async / altro_mpc_run_async) every step is then a one-step solve kernel followed by the polish kernel:
   * the next step shifts the polished trajectory and the projected multipliers.
   * PARITY UNPINNED: the reference stores no polished trajectory. */
  int32_t projected_newton;
  double projected_newton_tolerance;   /* 1e-3 */
  double active_set_tolerance_pn;      /* 1e-3: inequality rows with c >= -tol join the polish's active set */
  double rho_chol;                     /* 1e-2: S + rho I is factored, the solve is refined against S */
  double rho_primal;                   /* 1e-8: added to the diagonal cost Hessian */
  double r_threshold;                  /* 1.1 */
} altro_opts;

#define ALTRO_TRACE_LEN 16 /* per-instance trace depth kept on device */

typedef struct altro_handle altro_handle;

/* SolverOptions() defaults */
int32_t altro_default_opts(altro_opts* opts);

/* ALTROSolver(prob, opts): random_linear_problem.jl:87, simple_rocket.jl:128,
 * grasp_mpc.jl:35, ALTROParams.jl:96.  Allocates every device workspace. */
int32_t altro_batch_create(const altro_dims* dims, const altro_opts* opts, int32_t device,
                           altro_handle** out);
int32_t altro_batch_destroy(altro_handle* h);
const char* altro_last_error(const altro_handle* h);

/* RD.LinearModel(A, B[, d]; dt): random_linear_problem.jl:8; LTV/affine: ALTROParams.jl:61,
 * linearized_dynamics.jl:69-96.   x+ = A x + B u + f.
 *   per_instance != 0: arrays hold `batch` blocks, else one block shared by all instances
 *   per_knot     != 0: each block holds N-1 knot blocks (LTV); for an (n, m) of the 16-lane kernel set this
 *                      must be the first call after create (the handle then moves to the wide kernel)
 * f may be NULL (zero). */
int32_t altro_batch_set_dynamics(altro_handle* h, const double* A, const double* B, const double* f,
                                 int32_t per_knot, int32_t per_instance);

/* TO.TrackingObjective(Q, R, Z; Qf) / LQRObjective with diagonal weights: mpc.jl:26-29.
 * Stage costs are scaled by dt, the terminal cost is not (random_linear_problem.jl:52-53). */
int32_t altro_batch_set_tracking_cost(altro_handle* h, const double* Qdiag, const double* Rdiag,
                                      const double* Qfdiag, double dt);

/* add_constraint!(cons, con, inds): random_linear_problem.jl:23-24 (BoundConstraint),
 * rocket_landing_problem.jl:96,123-124,142,165, grasp_problem.jl:29-67, ALTROParams.jl:67-78.
 *   k_first..k_last: 0-based inclusive knot range (knot N-1 is terminal: state columns only)
 *   BOX:    zmin, zmax of length n+m (+-inf = absent)
 *   LINEAR: A [p][n+m] ROW-major, b [p]; value A z + b {= 0 | <= 0}
 *   SOC:    A, b as above; value v = A z + b with ||v[0..p-2]|| <= v[p-1]
 *   per_knot: bit 0 set: A, b hold one block per knot of the range (grasp_problem.jl:35-67);
 *             bit 1 set (values 2, 3): A, b hold those blocks once PER INSTANCE ([batch][knots][p][n+m] and
 *             [batch][knots][p]): every problem of the batch owns its tables, as the reference's problems do
 *             when mpc_update! rewrites them in place (grasp_mpc_helpers.jl:46-55) -- instances may then sit at
 *             different MPC steps or footholds.  The structure (kind, sense, knot range, p) stays common.
 * Without bit 1 the data is shared by all instances of the batch.  HIP library limits: one BOX; cones of
 * dimension 2..4; on the 16-lane kernels ((n,m) in (12,4) (12,3) (8,4) (6,6) (6,3)) at most
 * 16 LINEAR / SOC rows are active at any one knot (a cone of dimension 2..4 takes the first lanes
 * of an aligned group of 4, linear rows take any free lane; constraints with disjoint knot
 * ranges share lanes); every other (n <= 64, m <= 32) runs on the one-wave-per-instance kernel with
 * up to 64 LINEAR / SOC rows in total; constraints are added before the first solve. */
int32_t altro_batch_add_constraint(altro_handle* h, int32_t kind, int32_t sense, int32_t k_first,
                                   int32_t k_last, int32_t p, const double* A, const double* b,
                                   const double* zmin, const double* zmax, int32_t per_knot,
                                   int32_t* con_id);
/* in-place mutation of per-knot constraint data: grasp_mpc_helpers.jl:46-55.  A, b in the layout the
 * constraint was added with (per instance if it was); either may be NULL (unchanged). */
int32_t altro_batch_update_constraint_data(altro_handle* h, int32_t con_id, const double* A,
                                           const double* b);

/* TO.set_initial_state!: random_linear_problem.jl:130.  x0 is [batch][n]. */
int32_t altro_batch_set_initial_state(altro_handle* h, const double* x0);
/* TO.update_trajectory!(obj, Z_track, k): random_linear_problem.jl:133.
 * Xref [batch][N][n], Uref [batch][N-1][m]. */
int32_t altro_batch_set_reference(altro_handle* h, const double* Xref, const double* Uref);
/* initial_trajectory! / initial_controls! / initial_states!: mpc.jl:45, altro_solver.jl:70-71.
 * X may be NULL (iLQR re-rolls the states out from x0). */
int32_t altro_batch_set_initial_trajectory(altro_handle* h, const double* X, const double* U);
/* RD.shift_fill!(Z) and Altro.shift_fill!(conSet): random_linear_problem.jl:136,139 */
int32_t altro_batch_shift_fill(altro_handle* h, int32_t shift_primal, int32_t shift_dual);
/* set_options!: flexible_sat_mpc.jl:163 */
int32_t altro_batch_set_options(altro_handle* h, const altro_opts* opts);

/* solve!(solver): random_linear_problem.jl:113.  Synchronous on return. */
int32_t altro_batch_solve(altro_handle* h);
/* Same, but only enqueued on the handle's stream; pair with altro_batch_synchronize. */
int32_t altro_batch_solve_async(altro_handle* h);
int32_t altro_batch_synchronize(altro_handle* h);

/* benchmark_solve!(solver; samples, evals): random_linear_problem.jl:161 (samples = 5, evals = 5),
 * simple_rocket.jl:171, run_simple_rocket.jl:67,102, flexible_sat_mpc.jl:166.  Altro.jl's harness
 * around BenchmarkTools: Z0 = copy of the solver's trajectory; then 1 warm-up evaluation and
 * samples x evals timed evaluations of { initial_trajectory!(solver, Z0); solve!(solver) }.
 * Only the primal trajectory is restored: with reset_duals = false (run_random_linear.jl:47) every
 * repetition starts from the multipliers the previous one left, which is what the iteration counts
 * and times stored in the reference's *.jld2 files measure (the statistics of the LAST repetition;
 * random_linear_problem.jl:171-173).  On return the handle holds the result of the last repetition.
 * sample_ms (may be NULL) receives `samples` values: device time of one sample divided by evals,
 * i.e. BenchmarkTools' per-evaluation time of each sample, in ms for the whole batch.  Synchronous. */
int32_t altro_batch_benchmark_solve(altro_handle* h, int32_t samples, int32_t evals, float* sample_ms);

/* states(solver), controls(solver), Altro.get_duals: random_linear_problem.jl:177-181 */
int32_t altro_batch_get_states(altro_handle* h, double* X);
int32_t altro_batch_get_controls(altro_handle* h, double* U);
/* duals of one constraint: BOX -> [batch][nk][2][n+m] (upper rows, then lower rows; zero for
 * unbounded elements), LINEAR/SOC -> [batch][nk][p] */
int32_t altro_batch_get_duals(altro_handle* h, int32_t con_id, double* lambda);
int32_t altro_batch_set_duals(altro_handle* h, int32_t con_id, const double* lambda);

/* iterations(solver), status(solver), cost(solver), max_violation(solver), solver.stats:
 * random_linear_problem.jl:166-174.  Any output pointer may be NULL.  Arrays have `batch`
 * entries; traces are [batch][ALTRO_TRACE_LEN] (cost and max violation after each of the
 * first ALTRO_TRACE_LEN iLQR iterations). */
int32_t altro_batch_get_stats(altro_handle* h, int32_t* iterations, int32_t* iterations_outer,
                              int32_t* status, double* cost, double* c_max, double* cost_trace,
                              double* cmax_trace);
/* accepted line-search step of each of the first ALTRO_TRACE_LEN iLQR iterations of the last
 * solve, [batch][ALTRO_TRACE_LEN] (0 = the search failed and the trajectory was kept) */
int32_t altro_batch_get_alpha_trace(altro_handle* h, double* alpha_trace);
/* feedback gains K [batch][N-1] blocks of m x n (column-major) and feedforward d [batch][N-1][m]
 * left by the last backward pass of the last solve (backwardpass!, ilqr K/d; either may be NULL) */
int32_t altro_batch_get_gains(altro_handle* h, double* K, double* d);
/* device time of the last solve launch sequence on the handle's stream, HIP events (ms) */
int32_t altro_batch_last_solve_ms(altro_handle* h, float* ms);
/* Launch-duration history of the solve kernel (HIP events recorded on the handle's stream around
 * every solve launch since the last reset): the measurement behind bench.py's roofline figure.
 * reset also clears the work counters below.  Synchronises the stream.  timing_get returns the most recent
 * launches, at most 1024 of them (launch_ring.h CAP); the work counters keep accumulating over ALL launches since the
 * reset, so a caller that relates the two (bench.py) must stay within 1024 launches per reset (it asserts so). */
int32_t altro_batch_timing_reset(altro_handle* h);
int32_t altro_batch_timing_get(altro_handle* h, float* ms, int32_t capacity, int32_t* count);
/* Work done since the last timing reset, per instance: iLQR backward passes, rollouts (open-loop
 * + the alpha = 1 trial of every line search) and further line-search trials (evaluated without a
 * rollout, DESIGN.md "Line search").  These are the measured counts SURVEY 8(d)'s flops_solve
 * formula is evaluated with.  Arrays of `batch` int64; any pointer may be NULL. */
int32_t altro_batch_get_work_counters(altro_handle* h, int64_t* backward_passes, int64_t* rollouts,
                                      int64_t* trials);
/* Per instance since the last timing reset: solves run, iLQR iterations, solves that ended
 * SOLVE_SUCCEEDED.  Arrays of `batch` int64; any pointer may be NULL. */
int32_t altro_batch_get_solve_counters(altro_handle* h, int64_t* solves, int64_t* iterations, int64_t* succeeded);
/* Per instance since the last timing reset: iLQR iterations of the default (non-strict) mode that were confirmed
 * as converged by the first-order costate sweep alone -- no backward pass, no first-order sweep with the stored gains,
 * no rollout (altro_opts.strict).  Every iteration is exactly one of three kinds:
 * iterations = backward_passes + reused (altro_batch_get_reuse_counter) + confirmed. */
int32_t altro_batch_get_confirm_counter(altro_handle* h, int64_t* confirmed);
/* Per instance since the last timing reset: iLQR iterations of the default mode that took their gains from memory
 * instead of running a backward pass -- the active set (hashed knot by knot) and the penalty were those of the pass
 * that left the gains there, in this solve or an earlier one, and inside a fixed active set K and Quu do not depend
 * on the trajectory.  Such an iteration runs the first-order recursion (d_k = -Quu^-1 Qu, s_k = Qx + K' Qu, dV) and
 * then its rollout as usual; counted in `iterations`, not in `backward_passes` (altro_opts.strict = 1: never taken).
 * Every setter the gains depend on (dynamics, cost, constraints, options) drops them. */
int32_t altro_batch_get_reuse_counter(altro_handle* h, int64_t* reused);
/* Projected-Newton polish of the last solve, per instance (arrays of `batch`; any pointer may be NULL): ran (the AL
 * stage ended SOLVE_SUCCEEDED-or-unsolved above constraint_tolerance), failed (a block of D H^-1 D' + rho I was not
 * positive definite), residual (final max |d| over the active rows, the initial condition and the dynamics defects).
 * All zero when altro_opts.projected_newton = 0. */
int32_t altro_batch_get_polish_stats(altro_handle* h, int32_t* ran, int32_t* failed, double* residual);
/* The dual half of the polish (multiplier_projection! of Altro.jl's ProjectedNewtonSolver; ALTRO, IROS 2019, IV-B): at
 * the polished trajectory the multipliers of the polish's active rows D (initial condition, active constraint rows,
 * dynamics) are projected, lam <- lam - (D D')^-1 D (g + D' lam), g the gradient of the cost.  Per instance (arrays of
 * `batch`; any pointer may be NULL): the stationarity residual ||g + D' lam||_2 `before` (AL duals on the active box /
 * linear rows, zero elsewhere) and `after` the projection, and whether a block of D D' was not positive definite.  The
 * projected multipliers stay inside the polish, as Altro's do; the AL duals (altro_batch_get_duals) are untouched.
 * All zero when the polish did not run. */
int32_t altro_batch_get_polish_dual_residuals(altro_handle* h, double* before, double* after, int32_t* failed);
/* Diagnostic (16-lane kernels): 16 int64 per wave (4 instances) of the last solve launch.  [0] s_memtime ticks in
 * total; [7] backward passes the wave ran in the lone-row form (one instance over the four DPP rows).  The
 * -DALTRO_PHASE_STAMPS build also fills ticks per phase -- [1] four-row backward passes, [2] closed-loop rollouts,
 * [3] open-loop rollouts, [4] Todorov gradient, [5] dual update, [6] line-search sweeps, [8] lone-row backward passes,
 * [9] first-order sweeps, [10] costate sweeps -- and how many of each the wave ran: [11] four-row passes, [12]
 * first-order sweeps, [13] costate sweeps, [14] closed-loop rollouts, [15] trial sweeps.  count = 16 * waves. */
int32_t altro_batch_get_wave_cycles(altro_handle* h, int64_t* cycles, int32_t capacity, int32_t* count);

/* ---- device-resident MPC harness (reference random_linear_problem.jl:121-139, mpc.jl:11-47).
 * The reference's MPC loop runs on the host around solve!; for a batch that lives in HBM the
 * same update sequence is provided on device so that no step crosses PCIe. */

/* Long reference trajectory Z_track (run_random_linear.jl:111-112): Xtrack [batch][Nt][n],
 * Utrack [batch][Nt-1][m].  Also installs window 0 as reference and initial trajectory
 * (gen_tracking_problem, mpc.jl:19-45). */
int32_t altro_mpc_set_track(altro_handle* h, const double* Xtrack, const double* Utrack, int32_t Nt);
/* unit-normal samples for the 1 % plant noise (random_linear_problem.jl:129): [steps][batch][n] */
int32_t altro_mpc_set_noise(altro_handle* h, const double* noise, int32_t steps);
/* Plant-noise model of the device-side MPC step: x0_i += noise_i * weights[i] * norm, with
 *   mode 0: norm = ||x0||_inf over all states (random_linear_problem.jl:129; default, weights 1/100)
 *   mode 1: norm = ||x0[group_i]||_2, groups[i] in {0,1} (simple_rocket.jl:65-71: positions with
 *           weight 1/1000, velocities with weight 1/100)
 *   mode 2: norm = 1, absolute noise (flexible_sat_mpc.jl:266: 0.0002 * randn) */
int32_t altro_mpc_set_noise_model(altro_handle* h, int32_t mode, const double* weights, const int32_t* groups);
/* shift = 0: the device MPC step keeps the previous solution and duals as the warm start instead of
 * shifting them by one knot (flexible_sat_mpc.jl:275-276 leaves both shift_fill! calls commented
 * out); default 1 */
int32_t altro_mpc_set_shift(altro_handle* h, int32_t shift);
/* Per-knot (LTV) dynamics for the device-resident MPC loop.  The reference re-linearises its model before
 * every solve (update_dynamics_matrices!, altro_solver.jl:5-37: model.A[k], B[k], d[k] for the knots of the
 * new horizon); for a loop that runs on the device the blocks of every step are uploaded once.  Each
 * instance (or all of them, per_instance = 0) owns `nblocks` knot blocks A [n x n], B [n x m], f [n]
 * (column-major, f may be NULL); the solve whose reference window starts at knot r -- MPC step i has
 * r = i + 1, a plain solve before the loop r = 0 -- reads block r * step_stride + k for its knot k:
 *   step_stride = 1      blocks indexed by absolute knot, like the reference track (a linearisation that
 *                        depends on time only: gait schedule, planned footholds); nblocks >= steps + N
 *   step_stride = N - 1  one full table of N - 1 blocks per step (re-linearisation about anything)
 * The plant step of altro_mpc_step_async uses knot 0 of the window that is current before the step.
 * Replaces altro_batch_set_dynamics; like it, must be the first call on an (n, m) of the 16-lane set. */
int32_t altro_mpc_set_dynamics_track(altro_handle* h, const double* A, const double* B, const double* f,
                                     int32_t nblocks, int32_t step_stride, int32_t per_instance);
/* One MPC step i (0-based), enqueued on the handle's stream, in the reference's order
 * (random_linear_problem.jl:125-139,161): x0 <- A x_1 + B u_1 + noise_i*||.||_inf/100;
 * reference window <- i+1; primal shift_fill; dual shift_fill; solve. */
int32_t altro_mpc_step_async(altro_handle* h, int32_t step);
/* The first half of that sequence only, for harnesses that keep the reference's own call order
 * around benchmark_solve! (random_linear_problem.jl:125-133): x0 <- A x_1 + B u_1 + noise_step*...,
 * reference window <- step+1.  No shift_fill and no solve: follow with altro_batch_shift_fill and
 * altro_batch_solve / altro_batch_benchmark_solve. */
int32_t altro_mpc_prepare_async(altro_handle* h, int32_t step);
/* The same for `nsteps` consecutive steps first_step .. first_step+nsteps-1 in ONE launch.
 * Instances are independent closed loops, so inside the launch each wavefront runs its own four
 * instances through all the steps without waiting for the rest of the batch; results are
 * bit-identical to nsteps calls of altro_mpc_step_async.  Per-step statistics are accumulated
 * (altro_batch_get_solve_counters); altro_batch_get_stats reports the last step. */
int32_t altro_mpc_run_async(altro_handle* h, int32_t first_step, int32_t nsteps);
/* x0 currently installed: [batch][n] */
int32_t altro_batch_get_initial_state(altro_handle* h, double* x0);

/* Diagnostic switches (no Julia counterpart: Altro.jl has no scheduling to switch).  The library reads nothing from the
 * environment; a test or a measuring tool that wants one of the kernels' scheduling features off -- to show that it
 * changes no result, or to time it -- says so here.  h == NULL: for the handles THIS THREAD creates afterwards;
 * otherwise for that handle, from its next launch on.  Keys (value 0 restores the default):
 *   "no_lone", "no_shadow", "no_resync", "no_group", "no_reuse", "no_qz_pass", "no_mate_rank"   one scheduling feature of the 16-lane kernels off
 *   "group_mode" 0..4, "group_max_steps", "trace_wave"              slot order of a grouped launch / diagnostic builds
 *   "force_wide", "wide_compact", "wide_coop", "wide_static_mask"   read at altro_batch_create: NULL handle only
 *   "keep_gains"   the setters stop dropping the stored gains (the product then returns results from STALE gains: it
 *                  exists to show that the tests notice).  Compiled into -DALTRO_DEBUG builds only; a release build
 *                  answers ALTRO_ERR_UNSUPPORTED.
 * Unknown key: ALTRO_ERR_INVALID_ARG. */
int32_t altro_debug_set(altro_handle* h, const char* key, int32_t value);

/* the handle's hipStream_t, for callers that order their own device work after a solve */
int32_t altro_batch_get_stream(altro_handle* h, void** stream);

#ifdef __cplusplus
}
#endif
#endif
