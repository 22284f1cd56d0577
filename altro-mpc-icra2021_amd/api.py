"""Host-side mirror of the Altro.jl / TrajectoryOptimization.jl / RobotDynamics.jl calls the
reference's benchmark scripts make, for a BATCH of independent instances solved on MI355X.

Julia name (reference call site)                      -> here
  RD.LinearModel(A, B[, d]; dt)   (random_linear_problem.jl:8)  -> LinearModel
  TO.TrackingObjective / LQRObjective (mpc.jl:29)      -> TrackingObjective
  BoundConstraint(n,m,u_min,u_max) (random_linear_problem.jl:23) -> BoundConstraint
  ConstraintList / add_constraint! (random_linear_problem.jl:22-24) -> ConstraintList.add_constraint
  Problem(model,obj,xf,tf;x0,constraints) (mpc.jl:42)  -> Problem
  SolverOptions(...) / set_options! (run_random_linear.jl:41-49) -> SolverOptions / set_options
  ALTROSolver(prob, opts) (random_linear_problem.jl:87) -> ALTROSolver
  solve!(altro) (:113)                                  -> solve(altro)
  TO.set_initial_state! (:130)                          -> set_initial_state(altro, x0)
  TO.update_trajectory!(obj, Z_track, k) (:133)         -> update_trajectory(altro, Xref, Uref)
  RD.shift_fill!(Z) (:136), Altro.shift_fill!(conSet) (:139) -> shift_fill(altro, primal, dual)
  states / controls / iterations / status / cost / max_violation (:166-181) -> same names
  benchmark_solve!(altro; samples, evals) (:161)        -> benchmark_solve(altro, ...)

Arrays are numpy, instance-major: X (B, N, n), U (B, N-1, m), A (B, n, n) in natural
(row, col) indexing; conversion to the C-ABI's column-major blocks happens here.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _lib
from ._lib import AltroError, SOLVE_SUCCEEDED, STATUS_NAMES  # noqa: F401

_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int32)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(_DP)


def SolverOptions(**kw):
    """Altro.SolverOptions with Altro.jl's defaults; keyword names as in the reference."""
    o = _lib.Opts()
    rc = _lib.lib().altro_default_opts(C.byref(o))
    if rc:
        raise AltroError(rc, "altro_default_opts")
    for k, v in kw.items():
        if k in ("verbose", "show_summary", "static_bp", "save_S"):
            # accepted for source compatibility: they only affect printing or Julia-side memory layout
            continue
        if not hasattr(o, k):
            raise KeyError(f"unknown SolverOptions field {k}")
        setattr(o, k, v)
    return o


@dataclass
class LinearModel:
    """RD.LinearModel: x+ = A x + B u (+ d).  A: (n,n) shared or (B,n,n) per instance; with
    per_knot=True (a model built with `times`, ALTROParams.jl:61) one block per knot:
    (N-1,n,n) or (B,N-1,n,n), d likewise."""
    A: np.ndarray
    B: np.ndarray
    d: Optional[np.ndarray] = None
    dt: float = 0.1
    per_knot: bool = False


@dataclass
class TrackingObjective:
    """Diagonal tracking cost about (Xref, Uref); stage costs are scaled by dt."""
    Q: np.ndarray
    R: np.ndarray
    Qf: np.ndarray
    Xref: np.ndarray   # (B, N, n)
    Uref: np.ndarray   # (B, N-1, m)


@dataclass
class BoundConstraint:
    n: int
    m: int
    x_min: Optional[np.ndarray] = None
    x_max: Optional[np.ndarray] = None
    u_min: Optional[np.ndarray] = None
    u_max: Optional[np.ndarray] = None

    def zbounds(self):
        def full(v, k, fill):
            if v is None:
                return np.full(k, fill)
            return np.broadcast_to(np.asarray(v, dtype=np.float64), (k,)).copy()
        zmin = np.r_[full(self.x_min, self.n, -np.inf), full(self.u_min, self.m, -np.inf)]
        zmax = np.r_[full(self.x_max, self.n, np.inf), full(self.u_max, self.m, np.inf)]
        return zmin, zmax


@dataclass
class LinearConstraint:
    """A z + b {= 0 | <= 0}; A is (p, n+m) on z = [x; u].  Mirrors TO.LinearConstraint /
    GoalConstraint(xf) (A = [I 0], b = -xf) / LinearizedFrictionConstraint."""
    A: np.ndarray
    b: np.ndarray
    equality: bool = False
    per_instance: bool = False   # A is (B, p, n+m) or (B, nk, p, n+m): every instance of the batch owns its data


@dataclass
class NormConstraint:
    """Second-order cone: ||(A z + b)[0:p-1]|| <= (A z + b)[p-1].  Mirrors NormConstraint(n, m, val,
    SecondOrderCone(), :control) (rocket_landing_problem.jl:123) and NormConstraint2 ([A y; c'y],
    new_constraints.jl:72-120) with their rows written on z."""
    A: np.ndarray
    b: np.ndarray
    per_instance: bool = False   # as LinearConstraint.per_instance


def GoalConstraint(xf, n, m):
    """GoalConstraint(xf) (rocket_landing_problem.jl:96): x_N = xf, added at knot N."""
    xf = np.asarray(xf, dtype=np.float64)
    return LinearConstraint(np.hstack([np.eye(n), np.zeros((n, m))]), -xf, equality=True)


@dataclass
class ConstraintList:
    n: int
    m: int
    N: int
    items: List = field(default_factory=list)

    def add_constraint(self, con, inds):
        """inds: 1-based inclusive range (first, last) as in Julia's `1:N-1`."""
        first, last = (inds.start, inds.stop - 1) if isinstance(inds, range) else inds
        self.items.append((con, int(first), int(last)))


@dataclass
class Problem:
    model: LinearModel
    obj: TrackingObjective
    constraints: ConstraintList
    x0: np.ndarray          # (B, n)
    N: int
    U0: Optional[np.ndarray] = None  # (B, N-1, m) initial controls; default: the reference controls

    @property
    def batch(self):
        return self.x0.shape[0]


class ALTROSolver:
    """ALTROSolver(prob, opts): owns a device-resident batch of solver workspaces."""

    def __init__(self, prob: Problem, opts=None, device=0):
        L = _lib.lib()
        self._L = L
        self.prob = prob
        B, n = prob.x0.shape
        m = np.asarray(prob.obj.R).shape[-1]
        self.B, self.n, self.m, self.N = B, n, m, prob.N
        self.opts = opts if opts is not None else SolverOptions()
        dims = _lib.Dims(B, n, m, prob.N)
        h = C.c_void_p()
        _lib.sync_debug_env()   # the tests' / tools' ALTRO_* switches reach the library through altro_debug_set, not getenv
        rc = L.altro_batch_create(C.byref(dims), C.byref(self.opts), device, C.byref(h))
        if rc:
            raise AltroError(rc, L.altro_last_error(None).decode())
        self.h = h
        self.con_ids = []
        mdl = prob.model
        set_dynamics(self, mdl)
        self._chk(L.altro_batch_set_tracking_cost(h, _p(_c(prob.obj.Q)), _p(_c(prob.obj.R)), _p(_c(prob.obj.Qf)), mdl.dt))
        for con, first, last in prob.constraints.items:
            if isinstance(con, BoundConstraint):
                zmin, zmax = con.zbounds()
                cid = C.c_int32(-1)
                self._chk(L.altro_batch_add_constraint(h, _lib.CON_BOX, _lib.SENSE_INEQ, first - 1, last - 1, 0,
                                                       None, None, _p(_c(zmin)), _p(_c(zmax)), 0, C.byref(cid)))
                self.con_ids.append(cid.value)
            elif isinstance(con, (LinearConstraint, NormConstraint)):
                A, b = _c(con.A), _c(con.b)
                per_inst = bool(getattr(con, "per_instance", False))
                per_knot = A.ndim == (4 if per_inst else 3)   # (nk, p, n+m): LinearConstraintTraj / AffineSOCTraj
                assert A.shape[-1] == n + m and A.shape[:-1] == b.shape
                assert not per_inst or A.shape[0] == B
                assert not per_knot or A.shape[-3] == last - first + 1
                soc = isinstance(con, NormConstraint)
                kind = _lib.CON_SOC if soc else _lib.CON_LINEAR
                sense = _lib.SENSE_EQ if (not soc and con.equality) else _lib.SENSE_INEQ
                cid = C.c_int32(-1)
                self._chk(L.altro_batch_add_constraint(h, kind, sense, first - 1, last - 1, A.shape[-2],
                                                       _p(A), _p(b), None, None, int(per_knot) | (2 if per_inst else 0), C.byref(cid)))
                self.con_ids.append(cid.value)
            else:
                raise AltroError(_lib.ERR_UNSUPPORTED, f"constraint type {type(con).__name__} is not built yet")
        update_trajectory(self, prob.obj.Xref, prob.obj.Uref)
        set_initial_state(self, prob.x0)
        U0 = prob.U0 if prob.U0 is not None else prob.obj.Uref
        initial_controls(self, U0)

    def _chk(self, rc):
        if rc:
            raise AltroError(rc, self._L.altro_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self._L.altro_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def set_dynamics(solver, mdl):
    """Install (or replace) the dynamics: the quadruped controller rewrites model.A[k], B[k], d[k]
    before every solve (altro_solver.jl:5-37)."""
    A = np.asarray(mdl.A, dtype=np.float64)
    Bm = np.asarray(mdl.B, dtype=np.float64)
    per_instance = A.ndim == (4 if mdl.per_knot else 3)
    Ac = _c(np.swapaxes(A, -1, -2))
    Bc = _c(np.swapaxes(Bm, -1, -2))
    dc = _c(mdl.d) if mdl.d is not None else None
    solver._chk(solver._L.altro_batch_set_dynamics(solver.h, _p(Ac), _p(Bc), _p(dc), int(mdl.per_knot), int(per_instance)))


def set_dynamics_track(solver, A, B, d=None, step_stride=1):
    """Per-knot dynamics of every MPC step, uploaded once for the device-resident loop
    (altro_mpc_set_dynamics_track; reference: update_dynamics_matrices!, altro_solver.jl:5-37).
    A: (nblocks, n, n) shared or (B, nblocks, n, n) per instance, B and d likewise; the solve of MPC
    step i reads block (i + 1) * step_stride + k for knot k (step_stride 1: blocks indexed by absolute
    knot; N - 1: one table per step)."""
    A = np.asarray(A, dtype=np.float64)
    Bm = np.asarray(B, dtype=np.float64)
    per_instance = A.ndim == 4
    nblocks = A.shape[-3]
    Ac = _c(np.swapaxes(A, -1, -2))
    Bc = _c(np.swapaxes(Bm, -1, -2))
    dc = _c(d) if d is not None else None
    solver._chk(solver._L.altro_mpc_set_dynamics_track(solver.h, _p(Ac), _p(Bc), _p(dc), int(nblocks), int(step_stride), int(per_instance)))


def set_tracking_cost(solver, Q, R, Qf, dt=None):
    """Replace the diagonal weights of the tracking objective (TO.TrackingObjective(Q, R, Z; Qf), mpc.jl:26-29)."""
    dt = solver.prob.model.dt if dt is None else dt
    solver._chk(solver._L.altro_batch_set_tracking_cost(solver.h, _p(_c(Q)), _p(_c(R)), _p(_c(Qf)), dt))


def set_options(solver, **kw):
    for k, v in kw.items():
        setattr(solver.opts, k, v)
    solver._chk(solver._L.altro_batch_set_options(solver.h, C.byref(solver.opts)))


def set_initial_state(solver, x0):
    x0 = _c(x0)
    assert x0.shape == (solver.B, solver.n)
    solver._chk(solver._L.altro_batch_set_initial_state(solver.h, _p(x0)))


def update_trajectory(solver, Xref, Uref):
    Xr, Ur = _c(Xref), _c(Uref)
    assert Xr.shape == (solver.B, solver.N, solver.n) and Ur.shape == (solver.B, solver.N - 1, solver.m)
    solver._chk(solver._L.altro_batch_set_reference(solver.h, _p(Xr), _p(Ur)))


def initial_controls(solver, U):
    U = _c(U)
    assert U.shape == (solver.B, solver.N - 1, solver.m)
    solver._chk(solver._L.altro_batch_set_initial_trajectory(solver.h, None, _p(U)))


def shift_fill(solver, primal=True, dual=True):
    solver._chk(solver._L.altro_batch_shift_fill(solver.h, int(primal), int(dual)))


def solve(solver):
    solver._chk(solver._L.altro_batch_solve(solver.h))
    return solver


def benchmark_solve(solver, samples=10, evals=10):
    """benchmark_solve!(solver; samples, evals) (random_linear_problem.jl:161 with samples=5, evals=5):
    the solver's trajectory is saved, then 1 warm-up + samples x evals repetitions of
    { initial_trajectory!(solver, Z0); solve!(solver) } run on the device.  Duals and penalties are
    NOT restored between repetitions (Altro.jl restores the primal trajectory only), so with
    reset_duals=false the statistics left behind -- the ones the reference stores in its *.jld2
    files -- are those of a solve from converged multipliers.  Returns the per-sample times in ms
    for the whole batch (BenchmarkTools' trial: time of a sample / evals)."""
    ms = np.zeros(samples, dtype=np.float32)
    solver._chk(solver._L.altro_batch_benchmark_solve(solver.h, int(samples), int(evals),
                                                      ms.ctypes.data_as(C.POINTER(C.c_float))))
    return ms


def states(solver):
    X = np.empty((solver.B, solver.N, solver.n))
    solver._chk(solver._L.altro_batch_get_states(solver.h, _p(X)))
    return X


def controls(solver):
    U = np.empty((solver.B, solver.N - 1, solver.m))
    solver._chk(solver._L.altro_batch_get_controls(solver.h, _p(U)))
    return U


def get_duals(solver, con=0):
    c, first, last = solver.prob.constraints.items[con]
    nk = last - first + 1
    if isinstance(c, BoundConstraint):
        lam = np.empty((solver.B, nk, 2, solver.n + solver.m))
    else:
        lam = np.empty((solver.B, nk, np.asarray(c.b).shape[-1]))
    solver._chk(solver._L.altro_batch_get_duals(solver.h, solver.con_ids[con], _p(lam)))
    return lam


def set_duals(solver, lam, con=0):
    lam = _c(lam)
    solver._chk(solver._L.altro_batch_set_duals(solver.h, solver.con_ids[con], _p(lam)))


@dataclass
class Stats:
    iterations: np.ndarray
    iterations_outer: np.ndarray
    status: np.ndarray
    cost: np.ndarray
    c_max: np.ndarray
    cost_trace: np.ndarray
    cmax_trace: np.ndarray
    tsolve_ms: float


def stats(solver):
    B = solver.B
    it = np.empty(B, dtype=np.int32)
    ito = np.empty(B, dtype=np.int32)
    st = np.empty(B, dtype=np.int32)
    cost_ = np.empty(B)
    cm = np.empty(B)
    jt = np.empty((B, _lib.TRACE_LEN))
    ct = np.empty((B, _lib.TRACE_LEN))
    solver._chk(solver._L.altro_batch_get_stats(
        solver.h, it.ctypes.data_as(_IP), ito.ctypes.data_as(_IP), st.ctypes.data_as(_IP),
        _p(cost_), _p(cm), _p(jt), _p(ct)))
    ms = C.c_float(0)
    rc = solver._L.altro_batch_last_solve_ms(solver.h, C.byref(ms))
    return Stats(it, ito, st, cost_, cm, jt, ct, ms.value if rc == 0 else float("nan"))


def iterations(solver):
    return stats(solver).iterations


def status(solver):
    return stats(solver).status


def cost(solver):
    return stats(solver).cost


def max_violation(solver):
    return stats(solver).c_max


def timing_reset(solver):
    solver._chk(solver._L.altro_batch_timing_reset(solver.h))


def timing_get(solver):
    """Durations (ms) of every solve-kernel launch since timing_reset, from HIP events recorded
    on the library's own stream."""
    cnt = C.c_int32(0)
    solver._chk(solver._L.altro_batch_timing_get(solver.h, None, 0, C.byref(cnt)))
    ms = np.zeros(cnt.value, dtype=np.float32)
    if cnt.value:
        solver._chk(solver._L.altro_batch_timing_get(solver.h, ms.ctypes.data_as(C.POINTER(C.c_float)), cnt.value, C.byref(cnt)))
    return ms


def work_counters(solver):
    """(backward passes, rollouts, interpolated line-search trials) per instance since
    timing_reset."""
    i64 = C.POINTER(C.c_int64)
    a = [np.zeros(solver.B, dtype=np.int64) for _ in range(3)]
    solver._chk(solver._L.altro_batch_get_work_counters(solver.h, *[x.ctypes.data_as(i64) for x in a]))
    return tuple(a)


def confirm_counter(solver):
    """iterations per instance (since timing_reset) that the default mode confirmed as converged with the
    first-order costate sweep instead of a backward pass (altro_batch_get_confirm_counter)."""
    a = np.zeros(solver.B, dtype=np.int64)
    solver._chk(solver._L.altro_batch_get_confirm_counter(solver.h, a.ctypes.data_as(C.POINTER(C.c_int64))))
    return a


def polish_stats(solver):
    """(ran, failed, residual) of the projected-Newton polish of the last solve, per instance
    (altro_batch_get_polish_stats; all zero with projected_newton = false)."""
    ran, failed = np.zeros(solver.B, dtype=np.int32), np.zeros(solver.B, dtype=np.int32)
    res = np.zeros(solver.B)
    ip = C.POINTER(C.c_int32)
    solver._chk(solver._L.altro_batch_get_polish_stats(solver.h, ran.ctypes.data_as(ip), failed.ctypes.data_as(ip), _p(res)))
    return ran, failed, res


def polish_dual_residuals(solver):
    """(before, after, failed) of the polish's multiplier projection, per instance: the stationarity residual
    ||g + D' lam||_2 with the AL duals and with the projected multipliers (altro_batch_get_polish_dual_residuals)."""
    a, b = np.zeros(solver.B), np.zeros(solver.B)
    f = np.zeros(solver.B, dtype=np.int32)
    solver._chk(solver._L.altro_batch_get_polish_dual_residuals(solver.h, _p(a), _p(b), f.ctypes.data_as(C.POINTER(C.c_int32))))
    return a, b, f


def reuse_counter(solver):
    """iterations per instance (since timing_reset) that took their gains from memory instead of running a backward
    pass (altro_batch_get_reuse_counter)."""
    a = np.zeros(solver.B, dtype=np.int64)
    solver._chk(solver._L.altro_batch_get_reuse_counter(solver.h, a.ctypes.data_as(C.POINTER(C.c_int64))))
    return a


def wave_cycles(solver):
    """(waves, 16) per wave of the last solve launch (16-lane kernels): s_memtime ticks in total (column 0) and, in
    the -DALTRO_PHASE_STAMPS build, per phase: 1 four-row backward passes, 2 closed-loop rollouts, 3 open-loop rollouts,
    4 Todorov gradient, 5 dual update, 6 line-search sweeps, 8 lone-row backward passes, 9 first-order sweeps,
    10 costate sweeps; 11-15 how many four-row passes, first-order sweeps, costate sweeps, closed-loop rollouts and
    trial sweeps the wave ran.  Column 7 (every build): backward passes run in the lone-row form."""
    cnt = C.c_int32(0)
    solver._chk(solver._L.altro_batch_get_wave_cycles(solver.h, None, 0, C.byref(cnt)))
    out = np.zeros(cnt.value, dtype=np.int64)
    solver._chk(solver._L.altro_batch_get_wave_cycles(solver.h, out.ctypes.data_as(C.POINTER(C.c_int64)), cnt.value, C.byref(cnt)))
    return out.reshape(-1, 16)


def solve_counters(solver):
    """(solves, iLQR iterations, SOLVE_SUCCEEDED count) per instance since timing_reset."""
    i64 = C.POINTER(C.c_int64)
    a = [np.zeros(solver.B, dtype=np.int64) for _ in range(3)]
    solver._chk(solver._L.altro_batch_get_solve_counters(solver.h, *[x.ctypes.data_as(i64) for x in a]))
    return tuple(a)


def update_constraint_data(solver, con, A=None, b=None):
    """In-place mutation of a constraint's (per-knot) data: grasp_mpc_helpers.jl:46-55."""
    solver._chk(solver._L.altro_batch_update_constraint_data(
        solver.h, solver.con_ids[con], _p(_c(A)) if A is not None else None, _p(_c(b)) if b is not None else None))


def alpha_trace(solver):
    """Accepted line-search step of the first TRACE_LEN iLQR iterations of the last solve."""
    a = np.empty((solver.B, _lib.TRACE_LEN))
    solver._chk(solver._L.altro_batch_get_alpha_trace(solver.h, _p(a)))
    return a


def gains(solver):
    """(K, d) of the last backward pass: K (B, N-1, m, n), d (B, N-1, m)."""
    K = np.empty((solver.B, solver.N - 1, solver.n, solver.m))
    d = np.empty((solver.B, solver.N - 1, solver.m))
    solver._chk(solver._L.altro_batch_get_gains(solver.h, _p(K), _p(d)))
    return np.swapaxes(K, -1, -2).copy(), d
