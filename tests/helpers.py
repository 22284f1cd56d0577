"""Shared test helpers: oracle set-up for the random-linear MPC workload and an independent
convex solve (condensed bounded least squares) used to pin the oracle's converged answers --
the reference's own validation method (ALTRO vs OSQP, random_linear_problem.jl:176-186)."""
import numpy as np
from scipy.optimize import lsq_linear

REF_OPTS = dict(cost_tolerance=1e-4, cost_tolerance_intermediate=1e-4, constraint_tolerance=1e-4,
                penalty_initial=1000.0, penalty_scaling=100.0, reset_duals=0)
"""run_random_linear.jl:41-49"""


def make_oracle(O, pb, b, opts=None, bounded=True):
    n, m, N = pb.n, pb.m, pb.N
    s = O.OracleSolver(n, m, N, pb.dt)
    s.set_dynamics(pb.A[b], pb.Bm[b])
    s.set_cost(np.full(n, pb.Qk), np.full(m, pb.Rk), np.full(n, pb.Qfk))
    if bounded:
        zmin = np.r_[np.full(n, -np.inf), np.full(m, -pb.u_bnd)]
        s.add_box(zmin, -zmin, 0, N - 2)
    s.set_opts(O.default_opts(**(opts or REF_OPTS)))
    Xr, Ur = pb.window(0)
    s.set_reference(Xr[b], Ur[b])
    s.set_initial_state(Xr[b, 0])
    s.set_controls(Ur[b])
    return s


def mpc_update(s, pb, b, i):
    """One pass of the reference MPC update order (random_linear_problem.jl:121-139)."""
    x0 = s.plant_step()
    x0 = x0 + pb.noise[i, b] * np.abs(x0).max() / 100.0
    s.set_initial_state(x0)
    Xr, Ur = pb.window(i + 1)
    s.set_reference(Xr[b], Ur[b])
    s.shift_fill(True, True)
    return x0


def condensed_qp(A, Bm, x0, Xref, Uref, Qd, Rd, Qfd, dt, u_bnd):
    """min_U sum dt(1/2|x-xr|_Q^2 + 1/2|u-ur|_R^2) + 1/2|x_N-xr_N|_Qf^2, |u|<=u_bnd, as a
    bounded least-squares problem in U (x eliminated through the dynamics)."""
    N, n = Xref.shape
    m = Uref.shape[1]
    nu = (N - 1) * m
    # X = Phi x0 + Gam U
    Phi = np.zeros((N * n, n))
    Gam = np.zeros((N * n, nu))
    Ak = np.eye(n)
    Phi[:n] = Ak
    for k in range(1, N):
        Gam[k * n:(k + 1) * n] = A @ Gam[(k - 1) * n:k * n]
        Gam[k * n:(k + 1) * n, (k - 1) * m:k * m] += Bm
        Ak = A @ Ak
        Phi[k * n:(k + 1) * n] = Ak
    wx = np.concatenate([np.sqrt(dt * np.asarray(Qd))] * (N - 1) + [np.sqrt(np.asarray(Qfd))])
    wu = np.concatenate([np.sqrt(dt * np.asarray(Rd))] * (N - 1))
    M = np.vstack([wx[:, None] * Gam, np.diag(wu)])
    rhs = np.concatenate([wx * (Xref.reshape(-1) - Phi @ x0), wu * Uref.reshape(-1)])
    res = lsq_linear(M, rhs, bounds=(-u_bnd, u_bnd), method="bvls", tol=1e-14, max_iter=2000)
    U = res.x.reshape(N - 1, m)
    X = (Phi @ x0 + Gam @ res.x).reshape(N, n)
    return X, U, res
