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


# ---------------------------------------------------------------------------------------------
# rocket landing (second-order cones + goal): oracle / GPU set-up shared by the tests
ROCKET_COLD_OPTS = dict(cost_tolerance_intermediate=1e-4, penalty_scaling=500.0, penalty_initial=1e-2,
                        constraint_tolerance=1e-5, iterations=5000, iterations_inner=100,
                        iterations_linesearch=100, iterations_outer=60)
"""run_simple_rocket.jl:39-50 (iterations_outer 500 there; 60 is never reached)"""
ROCKET_MPC_OPTS = dict(cost_tolerance=1e-4, cost_tolerance_intermediate=1e-4, constraint_tolerance=1e-4,
                       reset_duals=0, penalty_initial=1000.0, penalty_scaling=10.0)
"""run_simple_rocket.jl:121-129"""


def rocket_oracle(O, rp, x0, opts, Xref=None, Uref=None, U0=None, constraints=None):
    s = O.OracleSolver(rp.n, rp.m, rp.N, rp.dt)
    s.set_dynamics(rp.A, rp.Bm, rp.f)
    s.set_cost(rp.Q, rp.R, rp.Qf)
    s.set_reference(np.tile(rp.xf, (rp.N, 1)) if Xref is None else Xref,
                    np.zeros((rp.N - 1, rp.m)) if Uref is None else Uref)
    s.set_initial_state(x0)
    s.set_controls(rp.U0 if U0 is None else U0)
    s.con_ids = []
    for c in (rp.constraints if constraints is None else constraints):
        if c.A is None:        # a BOX spec
            s.con_ids.append(s.add_box(c.zmin, c.zmax, c.k_first, c.k_last))
        else:
            s.con_ids.append(s.add_affine(c.kind, c.sense, c.A, c.b, c.k_first, c.k_last))
    s.set_opts(O.default_opts(**opts))
    return s


def rocket_gpu_problem(altro, rp, x0, Xref=None, Uref=None, U0=None, constraints=None):
    return altro.mpc.constrained_problem(rp, x0, Xref, Uref, U0, constraints)


def soc_project(v):
    """Euclidean projection onto {(s, t): ||s|| <= t}."""
    s, t = v[:-1], v[-1]
    ns = np.linalg.norm(s)
    if ns <= t:
        return v.copy()
    if ns <= -t:
        return np.zeros_like(v)
    c = 0.5 * (1 + t / ns)
    return np.r_[c * s, c * ns]


def admm_conic_qp(Pm, q, G, h, cones, rho=1.0, sigma=1e-6, iters=20000, tol=1e-9):
    """min 1/2 x'Px + q'x  s.t.  G x + h in K,  K = product of cones [("zero"|"nonneg"|"soc", dim), ...].
    Operator-splitting (OSQP / COSMO form): an algorithm independent of AL-iLQR, used only to pin
    the oracle's converged conic solutions, as the reference pins ALTRO against COSMO / ECOS
    (simple_rocket.jl:183-203)."""
    nvar = Pm.shape[0]
    K = np.linalg.inv(Pm + sigma * np.eye(nvar) + rho * G.T @ G)
    x = np.zeros(nvar)
    z = np.zeros(G.shape[0])
    y = np.zeros(G.shape[0])

    def proj(w):   # projection onto C = K - h
        out = np.empty_like(w)
        i = 0
        for kind, d in cones:
            seg = w[i:i + d] + h[i:i + d]
            out[i:i + d] = (0.0 if kind == "zero" else np.maximum(seg, 0.0) if kind == "nonneg" else soc_project(seg)) - h[i:i + d]
            i += d
        return out

    for it in range(iters):
        x = K @ (sigma * x - q + G.T @ (rho * z - y))
        zt = G @ x
        zn = proj(zt + y / rho)
        y = y + rho * (zt - zn)
        rp_, rd_ = np.abs(zt - zn).max(), rho * np.abs(G.T @ (zn - z)).max()
        z = zn
        if it > 10 and rp_ < tol and rd_ < tol:
            break
    return x, it


# ---------------------------------------------------------------------------------------------
# quadruped (per-knot affine dynamics, friction pyramids, f_z box): oracle / GPU set-up
def quadruped_oracle(O, qp, x0, A, Bm, d, opts):
    s = O.OracleSolver(qp.n, qp.m, qp.N, qp.dt)
    s.set_dynamics(A, Bm, d)
    s.set_cost(qp.Q, qp.R, qp.Q)
    s.set_reference(np.tile(qp.x_des, (qp.N, 1)), np.zeros((qp.N - 1, qp.m)))
    s.set_initial_state(x0)
    s.set_controls(np.tile(qp.u_hover, (qp.N - 1, 1)))
    s.con_ids = []
    from altro_mpc_icra2021_amd import problems as P
    for c in qp.constraints:
        if c.kind == P.BOX:
            s.con_ids.append(s.add_box(c.zmin, c.zmax, c.k_first, c.k_last))
        else:
            s.con_ids.append(s.add_affine(c.kind, c.sense, c.A, c.b, c.k_first, c.k_last))
    s.set_opts(O.default_opts(**opts))
    return s


def quadruped_gpu_problem(altro, qp, x0, A, Bm, d):
    return altro.mpc.quadruped_problem(qp, x0, A, Bm, d)


def quadruped_condensed_qp(qp, x0, A, Bm, d):
    """The same problem as a dense QP in U for the ADMM solver: returns (P, q, G, h, cones, X(U))."""
    N, n, m = qp.N, qp.n, qp.m
    nu = (N - 1) * m
    Phi = np.zeros((N * n, n)); Gam = np.zeros((N * n, nu)); cvec = np.zeros(N * n)
    Phi[:n] = np.eye(n)
    for k in range(1, N):
        Phi[k * n:(k + 1) * n] = A[k - 1] @ Phi[(k - 1) * n:k * n]
        Gam[k * n:(k + 1) * n] = A[k - 1] @ Gam[(k - 1) * n:k * n]
        Gam[k * n:(k + 1) * n, (k - 1) * m:k * m] += Bm[k - 1]
        cvec[k * n:(k + 1) * n] = A[k - 1] @ cvec[(k - 1) * n:k * n] + d[k - 1]
    wx = np.concatenate([qp.dt * qp.Q] * (N - 1) + [qp.Q])
    wu = np.concatenate([qp.dt * qp.R] * (N - 1))
    base = Phi @ x0 + cvec - np.tile(qp.x_des, N)
    Pm = Gam.T @ (wx[:, None] * Gam) + np.diag(wu)
    q = Gam.T @ (wx * base)
    from altro_mpc_icra2021_amd import problems as P
    rows, hs = [], []
    for c in qp.constraints:
        for k in range(c.k_first, min(c.k_last, N - 2) + 1):
            if c.kind == P.BOX:
                for j in range(m):
                    e = np.zeros(nu); e[k * m + j] = 1.0
                    if np.isfinite(c.zmax[n + j]): rows.append(-e); hs.append(c.zmax[n + j])
                    if np.isfinite(c.zmin[n + j]): rows.append(e); hs.append(-c.zmin[n + j])
            else:
                for r in range(c.A.shape[0]):
                    e = np.zeros(nu); e[k * m:(k + 1) * m] = -c.A[r, n:]
                    assert not np.any(c.A[r, :n])
                    rows.append(e); hs.append(-c.b[r])
    G, h = np.array(rows), np.array(hs)
    return Pm, q, G, h, [("nonneg", len(hs))], (lambda U: (Phi @ x0 + Gam @ U + cvec).reshape(N, n))
