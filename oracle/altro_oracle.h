/*
 * altro_oracle.h -- CPU restatement of the ALTRO (AL-iLQR) solve used by the
 * benchmark scripts of RoboticExplorationLab/altro-mpc-icra2021.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and the cpu_baseline leg of bench.py may link or call it.  The
 * product path is the HIP library behind include/altro_batch.h and never routes here.
 *
 * Parity status: the solver arithmetic of the reference lives in un-vendored Julia
 * packages (Altro.jl 0.2.0 @socp, TrajectoryOptimization.jl 0.3.2 @socp,
 * RobotDynamics.jl 0.2.2; reference benchmarks/Manifest.toml:26-32,867-871,1030-1036)
 * and no Julia runtime exists in this pipeline.  ITERATION-LEVEL PARITY WITH Altro.jl IS
 * THEREFORE "PARITY UNPINNED".  What is pinned (tests/test_oracle_*.py):
 *   - converged solutions against an independent convex solve of the same problem (the
 *     reference's own validation method, random_linear_problem.jl:176-186),
 *   - the warm-start iteration-count statistics stored in the reference's
 *     horizon_comp.jld2 (median 2 / max 5; tests/golden/ref_iteration_stats.json),
 *   - the grasp cold-solve trajectory stored in grasp_ref_traj.jld2.
 *
 * Problem class (everything the reference's five benchmark problems need):
 *   dynamics   x_{k+1} = A_k x_k + B_k u_k + f_k           (LTI or per-knot; RD.LinearModel,
 *              reference random_linear_problem.jl:8, linearized_dynamics.jl:69-96)
 *   cost       sum_k dt*(1/2 dx'Q dx + 1/2 du'R du) + 1/2 dx_N' Qf dx_N, diagonal Q,R,Qf
 *              (TO.TrackingObjective / LQRObjective; reference mpc.jl:26-29)
 *   constraints, each on a knot range, value affine in z=[x;u]:
 *       BOX     z_min <= z <= z_max               (BoundConstraint, random_linear_problem.jl:23)
 *       LINEAR  A z + b  {=, <=} 0                (LinearConstraintTraj/GoalConstraint/
 *                                                  LinearizedFrictionConstraint)
 *       SOC     A z + b in second-order cone      (NormConstraint, NormConstraint2, AffineSOCTraj)
 *
 * All matrices crossing this API are COLUMN-MAJOR (Julia layout) unless noted.
 */
#ifndef ALTRO_ORACLE_H
#define ALTRO_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Altro.jl TerminationStatus (only UNSOLVED / SOLVE_SUCCEEDED are named in the reference:
 * simple_rocket.jl:144, random_linear_problem.jl:166). */
enum {
  ORC_UNSOLVED = 0, ORC_SOLVE_SUCCEEDED = 1, ORC_MAX_ITERATIONS = 2, ORC_MAX_ITERATIONS_OUTER = 3,
  ORC_MAXIMUM_COST = 4, ORC_STATE_LIMIT = 5, ORC_CONTROL_LIMIT = 6, ORC_NO_PROGRESS = 7,
  ORC_COST_INCREASE = 8
};

enum { ORC_BOX = 0, ORC_LINEAR = 1, ORC_SOC = 2 };
enum { ORC_EQ = 0, ORC_INEQ = 1 };

/* Altro.SolverOptions fields used by the reference (run_random_linear.jl:41-49,
 * run_simple_rocket.jl:39-50, grasp_benchmark.jl:19-34, ALTROParams.jl:86-95). */
typedef struct {
  double cost_tolerance;               /* 1e-4 */
  double cost_tolerance_intermediate;  /* 1e-4 */
  double gradient_tolerance;           /* 10   */
  double gradient_tolerance_intermediate; /* 1 */
  double constraint_tolerance;         /* 1e-6 */
  double penalty_initial;              /* NaN -> per-constraint default 1.0 */
  double penalty_scaling;              /* NaN -> per-constraint default 10  */
  double penalty_max;                  /* 1e8 */
  double dual_max;                     /* 1e8 */
  double line_search_lower_bound;      /* 1e-8 */
  double line_search_upper_bound;      /* 10 */
  double max_cost_value;               /* 1e8 */
  double max_state_value;              /* 1e8 */
  double max_control_value;            /* 1e8 */
  double bp_reg_initial;               /* 0 */
  double bp_reg_increase_factor;       /* 1.6 */
  double bp_reg_max;                   /* 1e8 */
  double bp_reg_min;                   /* 1e-8 */
  double bp_reg_fp;                    /* 10 */
  int iterations;                      /* 1000 */
  int iterations_inner;                /* 300 */
  int iterations_outer;                /* 30 */
  int iterations_linesearch;           /* 20 */
  int dJ_counter_limit;                /* 10 */
  int reset_duals;                     /* 1 */
  int reset_penalties;                 /* 1 */
  int bp_reg;                          /* 0 */
  int soc_second_order;                /* 1: add the projection-curvature term to the SOC Hessian */
  int kickout_max_penalty;             /* 0 (Altro.jl's default): the AL loop does NOT stop when the penalty reaches
                                          penalty_max, it goes on updating duals at the cap; 1: it stops there */
  /* projected-Newton polish (ALTRO, IROS 2019, Algorithm 4; Altro.jl solve!(::ALTROSolver)).  0 here: Altro.jl's own
   * default is true, but every script of the reference on this path sets it false; the one that does not
   * (old/altro_cold_solve.jl:79-86) ends with the polish skipped.  PARITY UNPINNED: nothing in the reference records a
   * trajectory the polish produced. */
  int projected_newton;
  double projected_newton_tolerance;   /* 1e-3: the AL stage stops at this violation, then the polish runs */
  double active_set_tolerance_pn;      /* 1e-3: an inequality row with c >= -tol is in the polish's active set */
  double rho_chol;                     /* 1e-2: S + rho I is what is factored (reg_solve refines against S) */
  double rho_primal;                   /* 1e-8: added to the (diagonal) cost Hessian */
  double r_threshold;                  /* 1.1: refinement stops when log(viol)/log(viol_prev) falls below it */
} orc_opts;

#define ORC_TRACE_MAX 256

typedef struct {
  int iterations;        /* total iLQR iterations   (Altro `iterations(solver)`) */
  int iterations_outer;  /* AL outer iterations */
  int status;
  double cost;           /* final AL cost J */
  double c_max;          /* final max violation */
  /* per inner iteration traces (first ORC_TRACE_MAX) */
  double J[ORC_TRACE_MAX];
  double dJ[ORC_TRACE_MAX];
  double grad[ORC_TRACE_MAX];
  double alpha[ORC_TRACE_MAX];
  double cmax_it[ORC_TRACE_MAX]; /* max violation of the accepted trajectory */
  /* per outer iteration */
  double c_max_outer[64];
  double penalty_max_outer[64];
  /* projected-Newton polish */
  int pn_ran, pn_failed;     /* the polish ran (AL ended above constraint_tolerance); a block of S was not positive definite */
  double pn_residual;        /* its final ||d||_inf (active rows, initial condition, dynamics defects) */
  /* multiplier projection after the primal polish: ||g + D' lam||_2 with the AL duals / with the projected multipliers */
  int pn_dual_failed;
  double pn_dual_residual0, pn_dual_residual;
} orc_stats;

typedef struct orc_solver orc_solver;

void orc_default_opts(orc_opts* o);

orc_solver* orc_create(int n, int m, int N, double dt);
void orc_destroy(orc_solver* s);

/* A: n*n col-major, B: n*m col-major, f: n (nullable).  per_knot!=0: arrays hold N-1 blocks. */
void orc_set_dynamics(orc_solver* s, const double* A, const double* B, const double* f, int per_knot);
void orc_set_cost(orc_solver* s, const double* Qd, const double* Rd, const double* Qfd);
/* Xref: N*n (knot-major), Uref: (N-1)*m */
void orc_set_reference(orc_solver* s, const double* Xref, const double* Uref);
void orc_set_initial_state(orc_solver* s, const double* x0);
void orc_set_controls(orc_solver* s, const double* U);
void orc_set_opts(orc_solver* s, const orc_opts* o);
int orc_debug_pass_trace(orc_solver* s, int* buf, int cap);
int orc_debug_ls_trace(orc_solver* s, double* buf, int cap);   /* diagnostic, see altro_oracle.c */

/* Returns constraint id.  k_first..k_last are 0-based inclusive knots (knot N-1 is terminal:
 * only state columns are used there).
 *   BOX:    zmin,zmax: n+m each (+-inf for absent); A,b ignored. rows = 2(n+m): [z-zmax ; zmin-z]
 *   LINEAR: A: p*(n+m) ROW-major per block, b: p; value c = A z + b; sense EQ or INEQ (c<=0)
 *   SOC:    same data; value v = A z + b must satisfy ||v[0:p-1]|| <= v[p-1]
 * per_knot!=0: A,b hold one block per knot in the range. */
int orc_add_constraint(orc_solver* s, int kind, int sense, int k_first, int k_last, int p,
                       const double* A, const double* b, const double* zmin, const double* zmax,
                       int per_knot);
void orc_update_constraint_data(orc_solver* s, int con, const double* A, const double* b);

/* RD.shift_fill!(Z) and Altro.shift_fill!(conSet) (random_linear_problem.jl:136,139) */
void orc_shift_fill(orc_solver* s, int primal, int dual);

/* solve!(::ALTROSolver) */
void orc_solve(orc_solver* s);

const double* orc_states(const orc_solver* s);    /* N*n */
const double* orc_controls(const orc_solver* s);  /* (N-1)*m */
const double* orc_gain_K(const orc_solver* s);    /* (N-1) blocks of m*n, column-major */
const double* orc_gain_d(const orc_solver* s);    /* (N-1)*m */
const orc_stats* orc_get_stats(const orc_solver* s);
int orc_num_duals(const orc_solver* s, int con);
const double* orc_duals(const orc_solver* s, int con);     /* [nk][p] */
const double* orc_penalties(const orc_solver* s, int con); /* [nk][p] */
void orc_set_duals(orc_solver* s, int con, const double* lam);

/* cost(solver) / max_violation(solver) evaluated at the current (X,U) after a fresh rollout */
double orc_cost(orc_solver* s);
double orc_max_violation(orc_solver* s);

/* discrete_dynamics at knot 0 with the current first control (random_linear_problem.jl:128) */
void orc_plant_step(const orc_solver* s, double* xnext);

#ifdef __cplusplus
}
#endif
#endif
