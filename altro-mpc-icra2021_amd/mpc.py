"""MPC harness for batches: restates benchmarks/mpc.jl::gen_tracking_problem (:11-47) and the
loop of benchmarks/random_linear_mpc/random_linear_problem.jl::run_MPC (:85-189) on top of the
batched solver.  The OSQP twin of the reference is replaced by the offline oracle in tests/.
"""
import ctypes as C

import numpy as np

from . import _lib, api

REF_OPTS = dict(cost_tolerance=1e-4, cost_tolerance_intermediate=1e-4, constraint_tolerance=1e-4,
                penalty_initial=1000.0, penalty_scaling=100.0, reset_duals=0)
"""SolverOptions of run_random_linear.jl:41-49 (projected_newton=false is the only built mode)."""


def gen_tracking_problem(pb, N=None):
    """gen_tracking_problem(prob, N): tracking cost Q=10 I, R=0.1 I, Qf=10 I about the first N
    knots of the long trajectory, same bound constraint on knots 1..N-1 (mpc.jl:11-47)."""
    N = pb.N if N is None else N
    n, m = pb.n, pb.m
    Xr, Ur = pb.Xtrack[:, :N], pb.Utrack[:, :N - 1]
    model = api.LinearModel(pb.A, pb.Bm, dt=pb.dt)
    obj = api.TrackingObjective(np.full(n, pb.Qk), np.full(m, pb.Rk), np.full(n, pb.Qfk), Xr, Ur)
    cons = api.ConstraintList(n, m, N)
    cons.add_constraint(api.BoundConstraint(n, m, u_min=-pb.u_bnd, u_max=pb.u_bnd), (1, N - 1))
    return api.Problem(model, obj, cons, x0=Xr[:, 0].copy(), N=N, U0=Ur.copy())


class BatchMPC:
    """Device-resident MPC loop over a RandomLinearBatch (reference run_MPC).

    The long reference trajectory and the per-step noise samples are uploaded once; every
    `step(i)` then runs entirely on the GPU in the reference's order:
    plant step + 1 % noise -> x0; retarget tracking cost to window i+1; primal shift_fill;
    dual shift_fill; solve  (random_linear_problem.jl:125-139,161).
    """

    def __init__(self, pb, opts=None, device=0):
        self.pb = pb
        self.solver = api.ALTROSolver(gen_tracking_problem(pb), opts or api.SolverOptions(**REF_OPTS), device)
        s = self.solver
        Xt, Ut = api._c(pb.Xtrack), api._c(pb.Utrack)
        s._chk(s._L.altro_mpc_set_track(s.h, api._p(Xt), api._p(Ut), pb.Nt))
        nz = api._c(pb.noise)
        s._chk(s._L.altro_mpc_set_noise(s.h, api._p(nz), nz.shape[0]))
        self.i = 0

    def initial_solve(self):
        """solve!(altro) before the loop (random_linear_problem.jl:113)."""
        api.solve(self.solver)

    def step_async(self, i=None):
        i = self.i if i is None else i
        s = self.solver
        s._chk(s._L.altro_mpc_step_async(s.h, i))
        self.i = i + 1

    def run_async(self, nsteps, first=None):
        """nsteps consecutive MPC steps in one launch (altro_mpc_run_async)."""
        first = self.i if first is None else first
        s = self.solver
        s._chk(s._L.altro_mpc_run_async(s.h, first, nsteps))
        self.i = first + nsteps

    def synchronize(self):
        s = self.solver
        s._chk(s._L.altro_batch_synchronize(s.h))

    def step(self, i=None):
        self.step_async(i)
        self.synchronize()

    def step_benchmark(self, i=None, samples=5, evals=5):
        """One MPC step exactly as the reference's loop body runs it (random_linear_problem.jl:121-161):
        plant step + noise -> x0; update_trajectory!; RD.shift_fill!(Z); Altro.shift_fill!(conSet);
        benchmark_solve!(altro, samples=5, evals=5).  The statistics read afterwards are those of the
        last of the 1 + samples*evals repeated solves, as in the reference's result Dict.  Returns the
        per-sample times (ms, whole batch)."""
        i = self.i if i is None else i
        s = self.solver
        s._chk(s._L.altro_mpc_prepare_async(s.h, i))
        api.shift_fill(s, True, True)
        ms = api.benchmark_solve(s, samples, evals)
        self.i = i + 1
        return ms

    def x0(self):
        s = self.solver
        out = np.empty((s.B, s.n))
        s._chk(s._L.altro_batch_get_initial_state(s.h, api._p(out)))
        return out


class TrackMPC:
    """Device-resident MPC loop for any tracking problem: the solver is built on the first window
    of (Xtrack, Utrack); `noise` are unit normals (steps, B, n).  noise_model: None for the
    random-linear model (1 % of ||x0||_inf), (weights, groups) for the two-group 2-norm model of
    the rocket benchmark (simple_rocket.jl:65-71), or (weights,) for absolute noise
    (flexible_sat_mpc.jl:266).  shift=False keeps the previous solution and duals as the warm
    start instead of shifting them (flexible_sat_mpc.jl:275-276)."""

    def __init__(self, prob, opts, Xtrack, Utrack, noise, noise_model=None, device=0, shift=True):
        self.solver = api.ALTROSolver(prob, opts, device)
        s = self.solver
        Xt, Ut = api._c(Xtrack), api._c(Utrack)
        s._chk(s._L.altro_mpc_set_track(s.h, api._p(Xt), api._p(Ut), Xt.shape[1]))
        api.set_initial_state(s, prob.x0)        # set_track starts from the track's first knot; the problem's x0 wins
        nz = api._c(noise)
        s._chk(s._L.altro_mpc_set_noise(s.h, api._p(nz), nz.shape[0]))
        if noise_model is not None:
            w = api._c(noise_model[0])
            if len(noise_model) > 1:
                g = np.ascontiguousarray(noise_model[1], dtype=np.int32)
                s._chk(s._L.altro_mpc_set_noise_model(s.h, 1, api._p(w), g.ctypes.data_as(C.POINTER(C.c_int32))))
            else:
                s._chk(s._L.altro_mpc_set_noise_model(s.h, 2, api._p(w), None))
        if not shift:
            s._chk(s._L.altro_mpc_set_shift(s.h, 0))
        self.i = 0

    initial_solve = BatchMPC.initial_solve
    step_async = BatchMPC.step_async
    run_async = BatchMPC.run_async
    synchronize = BatchMPC.synchronize
    step = BatchMPC.step
    step_benchmark = BatchMPC.step_benchmark
    x0 = BatchMPC.x0


def _add_specs(cons, specs, n, m):
    from . import problems as P
    for c in specs:
        if c.kind == P.BOX:
            cons.add_constraint(api.BoundConstraint(n, m, u_min=c.zmin[n:], u_max=c.zmax[n:]), (c.k_first + 1, c.k_last + 1))
        else:
            con = api.NormConstraint(c.A, c.b) if c.kind == P.SOC else api.LinearConstraint(c.A, c.b, equality=(c.sense == P.EQ))
            cons.add_constraint(con, (c.k_first + 1, c.k_last + 1))


def constrained_problem(data, x0, Xref=None, Uref=None, U0=None, constraints=None):
    """Problem(model, objective, ...; constraints) for the rocket / grasp data of problems.py
    (rocket_landing_problem.jl:66-186, grasp_problem.jl:1-107): x0 is (B, n); the reference
    defaults to the goal state, the initial controls to the data's guess."""
    B = x0.shape[0]
    model = api.LinearModel(data.A, data.Bm, data.f, dt=data.dt)
    Xr = np.tile(data.xf, (B, data.N, 1)) if Xref is None else Xref
    Ur = np.zeros((B, data.N - 1, data.m)) if Uref is None else Uref
    obj = api.TrackingObjective(data.Q, data.R, data.Qf, Xr, Ur)
    cons = api.ConstraintList(data.n, data.m, data.N)
    _add_specs(cons, data.constraints if constraints is None else constraints, data.n, data.m)
    return api.Problem(model, obj, cons, x0=x0, N=data.N, U0=np.tile(data.U0, (B, 1, 1)) if U0 is None else U0)


def quadruped_problem(qp, x0, A, Bm, d):
    """AltroParams (Structs/ALTROParams.jl:32-108) for a batch: x0 (B, 12); A, Bm, d per instance
    and per knot, (B, N-1, 12, 12) and (B, N-1, 12)."""
    B = x0.shape[0]
    model = api.LinearModel(A, Bm, d, dt=qp.dt, per_knot=True)
    obj = api.TrackingObjective(qp.Q, qp.R, qp.Q, np.tile(qp.x_des, (B, qp.N, 1)), np.zeros((B, qp.N - 1, qp.m)))
    cons = api.ConstraintList(qp.n, qp.m, qp.N)
    _add_specs(cons, qp.constraints, qp.n, qp.m)
    return api.Problem(model, obj, cons, x0=x0.copy(), N=qp.N, U0=np.tile(qp.u_hover, (B, qp.N - 1, 1)))
