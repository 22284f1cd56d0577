"""ctypes binding of the CPU oracle (oracle/altro_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libaltro_oracle.so")

TRACE_MAX = 256

BOX, LINEAR, SOC = 0, 1, 2
EQ, INEQ = 0, 1

STATUS_NAMES = ["UNSOLVED", "SOLVE_SUCCEEDED", "MAX_ITERATIONS", "MAX_ITERATIONS_OUTER",
                "MAXIMUM_COST", "STATE_LIMIT", "CONTROL_LIMIT", "NO_PROGRESS", "COST_INCREASE"]


class Opts(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        "cost_tolerance", "cost_tolerance_intermediate", "gradient_tolerance",
        "gradient_tolerance_intermediate", "constraint_tolerance", "penalty_initial",
        "penalty_scaling", "penalty_max", "dual_max", "line_search_lower_bound",
        "line_search_upper_bound", "max_cost_value", "max_state_value", "max_control_value",
        "bp_reg_initial", "bp_reg_increase_factor", "bp_reg_max", "bp_reg_min", "bp_reg_fp")] + \
        [(k, C.c_int) for k in (
            "iterations", "iterations_inner", "iterations_outer", "iterations_linesearch",
            "dJ_counter_limit", "reset_duals", "reset_penalties", "bp_reg", "soc_second_order",
            "kickout_max_penalty", "projected_newton")] + \
        [(k, C.c_double) for k in ("projected_newton_tolerance", "active_set_tolerance_pn", "rho_chol", "rho_primal", "r_threshold")]


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("iterations_outer", C.c_int), ("status", C.c_int),
                ("cost", C.c_double), ("c_max", C.c_double),
                ("J", C.c_double * TRACE_MAX), ("dJ", C.c_double * TRACE_MAX),
                ("grad", C.c_double * TRACE_MAX), ("alpha", C.c_double * TRACE_MAX),
                ("cmax_it", C.c_double * TRACE_MAX),
                ("c_max_outer", C.c_double * 64), ("penalty_max_outer", C.c_double * 64),
                ("pn_ran", C.c_int), ("pn_failed", C.c_int), ("pn_residual", C.c_double),
                ("pn_dual_failed", C.c_int), ("pn_dual_residual0", C.c_double), ("pn_dual_residual", C.c_double)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "altro_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_build/libaltro_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_default_opts.argtypes = [C.POINTER(Opts)]
        L.orc_set_dynamics.argtypes = [C.c_void_p, dp, dp, dp, C.c_int]
        L.orc_set_cost.argtypes = [C.c_void_p, dp, dp, dp]
        L.orc_set_reference.argtypes = [C.c_void_p, dp, dp]
        L.orc_set_initial_state.argtypes = [C.c_void_p, dp]
        L.orc_set_controls.argtypes = [C.c_void_p, dp]
        L.orc_set_opts.argtypes = [C.c_void_p, C.POINTER(Opts)]
        L.orc_add_constraint.restype = C.c_int
        L.orc_add_constraint.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         dp, dp, dp, dp, C.c_int]
        L.orc_update_constraint_data.argtypes = [C.c_void_p, C.c_int, dp, dp]
        L.orc_shift_fill.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_solve.argtypes = [C.c_void_p]
        L.orc_states.restype = dp
        L.orc_states.argtypes = [C.c_void_p]
        L.orc_controls.restype = dp
        L.orc_gain_K.restype = dp
        L.orc_gain_K.argtypes = [C.c_void_p]
        L.orc_gain_d.restype = dp
        L.orc_gain_d.argtypes = [C.c_void_p]
        L.orc_controls.argtypes = [C.c_void_p]
        L.orc_get_stats.restype = C.POINTER(Stats)
        L.orc_get_stats.argtypes = [C.c_void_p]
        L.orc_num_duals.restype = C.c_int
        L.orc_num_duals.argtypes = [C.c_void_p, C.c_int]
        L.orc_duals.restype = dp
        L.orc_duals.argtypes = [C.c_void_p, C.c_int]
        L.orc_penalties.restype = dp
        L.orc_penalties.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_duals.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_cost.restype = C.c_double
        L.orc_cost.argtypes = [C.c_void_p]
        L.orc_max_violation.restype = C.c_double
        L.orc_max_violation.argtypes = [C.c_void_p]
        L.orc_plant_step.argtypes = [C.c_void_p, dp]
        _lib = L
    return _lib


def _p(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_opts(**kw):
    o = Opts()
    lib().orc_default_opts(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    return o


class OracleSolver:
    """Single-instance AL-iLQR oracle.  Matrices are numpy arrays in natural (row, col)
    indexing; conversion to the column-major C layout happens here."""

    def __init__(self, n, m, N, dt):
        self.n, self.m, self.N, self.dt = n, m, N, dt
        self.h = lib().orc_create(n, m, N, dt)
        self.ncon = 0

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_destroy(self.h)
            self.h = None

    def set_dynamics(self, A, B, f=None):
        A = np.asarray(A, dtype=np.float64)
        B = np.asarray(B, dtype=np.float64)
        per_knot = A.ndim == 3
        # column-major blocks
        Ac = _c(np.swapaxes(A, -1, -2))
        Bc = _c(np.swapaxes(B, -1, -2))
        fc = _c(f) if f is not None else None
        lib().orc_set_dynamics(self.h, _p(Ac), _p(Bc), _p(fc), int(per_knot))

    def set_cost(self, Qd, Rd, Qfd):
        lib().orc_set_cost(self.h, _p(_c(Qd)), _p(_c(Rd)), _p(_c(Qfd)))

    def set_reference(self, Xref, Uref):
        Xr, Ur = _c(Xref), _c(Uref)
        assert Xr.shape == (self.N, self.n) and Ur.shape == (self.N - 1, self.m)
        lib().orc_set_reference(self.h, _p(Xr), _p(Ur))

    def set_initial_state(self, x0):
        lib().orc_set_initial_state(self.h, _p(_c(x0)))

    def set_controls(self, U):
        U = _c(U)
        assert U.shape == (self.N - 1, self.m)
        lib().orc_set_controls(self.h, _p(U))

    def set_opts(self, opts):
        lib().orc_set_opts(self.h, C.byref(opts))

    def add_box(self, zmin, zmax, k_first=0, k_last=None):
        k_last = self.N - 2 if k_last is None else k_last
        self.ncon += 1
        return lib().orc_add_constraint(self.h, BOX, INEQ, k_first, k_last, 0, None, None,
                                        _p(_c(zmin)), _p(_c(zmax)), 0)

    def add_affine(self, kind, sense, A, b, k_first, k_last):
        """A: (p, n+m) or (nk, p, n+m); b: (p,) or (nk, p).  value = A z + b."""
        A, b = _c(A), _c(b)
        per_knot = A.ndim == 3
        p = A.shape[-2]
        self.ncon += 1
        return lib().orc_add_constraint(self.h, kind, sense, k_first, k_last, p, _p(A), _p(b),
                                        None, None, int(per_knot))

    def update_constraint_data(self, con, A, b):
        lib().orc_update_constraint_data(self.h, con, _p(_c(A)) if A is not None else None,
                                         _p(_c(b)) if b is not None else None)

    def shift_fill(self, primal=True, dual=True):
        lib().orc_shift_fill(self.h, int(primal), int(dual))

    def solve(self):
        lib().orc_solve(self.h)
        return self.stats()

    def benchmark_solve(self, samples=5, evals=5):
        """Altro.jl's benchmark_solve!(solver; samples, evals) (reference call site
        random_linear_problem.jl:161): Z0 = copy of the trajectory; 1 warm-up + samples*evals
        repetitions of { initial_trajectory!(solver, Z0); solve! }.  Only the primal trajectory is
        restored (iLQR re-rolls the states out, so the controls are all that matters); duals and
        penalties carry over from one repetition to the next.  Returns the stats of the last one."""
        U0 = self.controls()
        st = None
        for _ in range(1 + samples * evals):
            self.set_controls(U0)
            st = self.solve()
        return st

    def stats(self):
        return lib().orc_get_stats(self.h).contents

    def states(self):
        return np.ctypeslib.as_array(lib().orc_states(self.h), shape=(self.N, self.n)).copy()

    def controls(self):
        return np.ctypeslib.as_array(lib().orc_controls(self.h), shape=(self.N - 1, self.m)).copy()

    def gains(self):
        """(K, d) of the last backward pass: K (N-1, m, n), d (N-1, m)."""
        K = np.ctypeslib.as_array(lib().orc_gain_K(self.h), shape=(self.N - 1, self.n, self.m)).copy()
        d = np.ctypeslib.as_array(lib().orc_gain_d(self.h), shape=(self.N - 1, self.m)).copy()
        return np.swapaxes(K, -1, -2).copy(), d

    def duals(self, con):
        k = lib().orc_num_duals(self.h, con)
        return np.ctypeslib.as_array(lib().orc_duals(self.h, con), shape=(k,)).copy()

    def penalties(self, con):
        k = lib().orc_num_duals(self.h, con)
        return np.ctypeslib.as_array(lib().orc_penalties(self.h, con), shape=(k,)).copy()

    def set_duals(self, con, lam):
        lam = _c(lam)
        assert lam.size == lib().orc_num_duals(self.h, con)
        lib().orc_set_duals(self.h, con, _p(lam))

    def cost(self):
        return lib().orc_cost(self.h)

    def max_violation(self):
        return lib().orc_max_violation(self.h)

    def plant_step(self):
        out = np.zeros(self.n)
        lib().orc_plant_step(self.h, _p(out))
        return out
