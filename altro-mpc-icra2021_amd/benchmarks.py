"""The reference's benchmark scripts, batched: each function runs one of them on the GPU for a
batch of independent problems and returns the result Dict of the reference (`:time`, `:iter`;
random_linear_problem.jl:188) as a Python dict of arrays, one column per instance.

  random_linear_mpc/run_random_linear.jl:110-153   -> run_random_linear, horizon / state / control sweeps
  rocket_landing/run_simple_rocket.jl:31-135        -> run_rocket
  grasp_optimization/grasp_benchmark.jl:60-85       -> run_grasp
  quadruped/Woofer/MPCControl/altro_solver.jl:40-88 -> run_quadruped

The OSQP / ECOS / COSMO twins of the reference are not part of this library (tests/ compare
against an offline oracle instead), so there is no `:err_traj` column.  Times are the kernel's
device time per MPC step for the whole batch (HIP events on the handle's stream)."""
import numpy as np

from . import api, mpc, problems

ROCKET_COLD_OPTS = dict(cost_tolerance_intermediate=1e-4, penalty_scaling=500.0, penalty_initial=1e-2,
                        constraint_tolerance=1e-5, iterations=5000, iterations_inner=100,
                        iterations_linesearch=100, iterations_outer=60)
"""run_simple_rocket.jl:39-50"""
ROCKET_MPC_OPTS = dict(cost_tolerance=1e-4, cost_tolerance_intermediate=1e-4, constraint_tolerance=1e-4,
                       reset_duals=0, penalty_initial=1000.0, penalty_scaling=10.0)
"""run_simple_rocket.jl:121-129"""
GRASP_COLD_OPTS = dict(cost_tolerance=1e-6, cost_tolerance_intermediate=1e-4, constraint_tolerance=1e-6,
                       iterations=5000, iterations_outer=60, iterations_inner=300)
"""grasp_benchmark.jl:19-25"""
GRASP_MPC_OPTS = dict(cost_tolerance=1e-4, cost_tolerance_intermediate=1e-3, constraint_tolerance=1e-4,
                      penalty_initial=10000.0, penalty_scaling=100.0)
"""grasp_benchmark.jl:26-34"""


def _result(times_ms, iters, ok, B):
    times_ms = np.asarray(times_ms)
    return {"time": times_ms, "time_us_per_solve": 1e3 * times_ms / B, "iter": np.asarray(iters),
            "solve_succeeded": np.asarray(ok), "batch": B}


def run_random_linear(n=12, m=4, N=50, batch=1024, steps=100, seed=1):
    """run_MPC(prob_mpc, opts, Z_track, 100) (random_linear_problem.jl:85-189)."""
    pb = problems.gen_random_linear_batch(batch, n=n, m=m, N=N, steps=steps, seed=seed)
    mp = mpc.BatchMPC(pb)
    mp.initial_solve()
    t, it, ok = [], [], []
    for i in range(steps):
        mp.step(i)
        st = api.stats(mp.solver)
        t.append(st.tsolve_ms); it.append(st.iterations.copy()); ok.append(st.status == api.SOLVE_SUCCEEDED)
    return _result(t, it, ok, batch)


def run_sweeps(batch=256, steps=100):
    """The three sweeps of run_random_linear.jl:110-153 (seeds 1, 10, 15 there)."""
    out = {"horizon": {}, "state_dim": {}, "control_dim": {}}
    for N in (11, 31, 51, 71, 101):
        out["horizon"][N] = run_random_linear(12, 6, N, batch, steps, seed=1)
    for n in (2, 15, 25, 35, 45, 55):
        out["state_dim"][n] = run_random_linear(n, 2, 21, batch, steps, seed=10)
    for m in (2, 6, 10, 15, 20, 25):
        out["control_dim"][m] = run_random_linear(30, m, 21, batch, steps, seed=15)
    return out


def run_rocket(batch=256, N_mpc=21, steps=100, N_cold=301, dt=0.05, seed=1):
    """Cold solve of the landing problem, then conic tracking MPC along it (run_simple_rocket.jl:31-135,
    simple_rocket.jl:59-82).  Instances differ in their initial state."""
    rp = problems.gen_rocket_problem(N=N_cold, tf=(N_cold - 1) * dt, Qfk=1e4, Rk=1.0, theta_thrust_max=5.0, theta_glideslope=45.0)
    rng = np.random.default_rng(seed)
    x0 = np.tile(rp.x0, (batch, 1)) + rng.standard_normal((batch, 6)) * np.array([1, 1, 1, .3, .3, .3]) * 0.5
    cold = api.ALTROSolver(mpc.constrained_problem(rp, x0), api.SolverOptions(**ROCKET_COLD_OPTS))
    api.solve(cold)
    cst = api.stats(cold)
    Xt, Ut = api.states(cold), api.controls(cold)
    cold.close()
    steps = min(steps, N_cold - N_mpc - 1)
    tp = problems.gen_rocket_problem(N=N_mpc, tf=dt * (N_mpc - 1), include_goal=False, theta_thrust_max=5.0, theta_glideslope=45.0)
    tp.Q, tp.R, tp.Qf = np.full(6, 10.0), np.full(3, 0.1), np.full(6, 10.0)            # gen_tracking_problem (mpc.jl:12-14)
    noise = rng.standard_normal((steps, batch, 6))
    prob = mpc.constrained_problem(tp, Xt[:, 0].copy(), Xt[:, :N_mpc].copy(), Ut[:, :N_mpc - 1].copy(), U0=Ut[:, :N_mpc - 1].copy())
    mp = mpc.TrackMPC(prob, api.SolverOptions(**ROCKET_MPC_OPTS), Xt, Ut, noise,
                      (np.array([1e-3] * 3 + [1e-2] * 3), np.array([0, 0, 0, 1, 1, 1])))
    mp.initial_solve()
    t, it, ok = [], [], []
    for i in range(steps):
        mp.step(i)
        st = api.stats(mp.solver)
        t.append(st.tsolve_ms); it.append(st.iterations.copy()); ok.append(st.status == api.SOLVE_SUCCEEDED)
    res = _result(t, it, ok, batch)
    res["cold"] = {"time": cst.tsolve_ms, "iter": cst.iterations, "solve_succeeded": cst.status == api.SOLVE_SUCCEEDED}
    return res


def run_grasp(batch=256, N_mpc=21, steps=30, N_cold=251, tf=25.0, seed=1):
    """Cold grasp solve, then run_grasp_mpc (grasp_mpc.jl:8-104): every step rewrites the per-knot
    constraint data of the shifted window (grasp_mpc_helpers.jl:1-55)."""
    import copy
    gp = problems.gen_grasp_problem(N=N_cold, tf=tf)
    x0c = np.tile(gp.x0, (batch, 1))
    cold = api.ALTROSolver(mpc.constrained_problem(gp, x0c), api.SolverOptions(**GRASP_COLD_OPTS))
    api.solve(cold)
    cst = api.stats(cold)
    Xt, Ut = api.states(cold)[0], api.controls(cold)[0]
    cold.close()
    steps = min(steps, N_cold - N_mpc - 1)

    def window(k0):
        return [problems.ConstraintSpec(c.kind, c.sense, 0, N_mpc - 2, A=c.A[k0:k0 + N_mpc - 1].copy(), b=c.b[k0:k0 + N_mpc - 1].copy())
                for c in gp.constraints[1:]]                                          # the goal is dropped (mpc.jl:33-40)
    tp = copy.copy(gp)
    tp.N, tp.Q, tp.R, tp.Qf = N_mpc, np.full(6, 1e3), np.full(6, 1.0), np.full(6, 10.0)   # grasp_benchmark.jl:79-80
    tp.constraints = window(0)
    Xr, Ur = np.tile(Xt[:N_mpc], (batch, 1, 1)), np.tile(Ut[:N_mpc - 1], (batch, 1, 1))
    sv = api.ALTROSolver(mpc.constrained_problem(tp, np.tile(Xt[0], (batch, 1)), Xr, Ur, U0=Ur.copy()), api.SolverOptions(**GRASP_MPC_OPTS))
    api.solve(sv)
    rng = np.random.default_rng(seed)
    t, it, ok = [], [], []
    for i in range(1, steps + 1):
        X, U = api.states(sv), api.controls(sv)
        xn = X[:, 0] @ gp.A.T + U[:, 0] @ gp.Bm.T + gp.f
        xn = xn + rng.standard_normal((batch, 6)) * np.abs(xn).max(axis=1, keepdims=True) / 100.0
        api.set_initial_state(sv, xn)
        api.update_trajectory(sv, np.tile(Xt[i:i + N_mpc], (batch, 1, 1)), np.tile(Ut[i:i + N_mpc - 1], (batch, 1, 1)))
        api.shift_fill(sv, True, False)
        for ci, c in enumerate(window(i)):
            api.update_constraint_data(sv, ci, c.A, c.b)
        api.shift_fill(sv, False, True)
        api.solve(sv)
        st = api.stats(sv)
        t.append(st.tsolve_ms); it.append(st.iterations.copy()); ok.append(st.status == api.SOLVE_SUCCEEDED)
    res = _result(t, it, ok, batch)
    res["cold"] = {"time": cst.tsolve_ms, "iter": cst.iterations[:1], "solve_succeeded": cst.status[:1] == api.SOLVE_SUCCEEDED}
    return res


def run_quadruped(batch=256, N=15, steps=30, linearized_friction=True, seed=7):
    """foot_forces! (altro_solver.jl:40-88) in a loop: re-linearise the per-knot dynamics for the
    advancing trot schedule, set x0, shift primal and dual, solve.  The plant here is the linear
    model's own first knot plus 1e-3 noise (the reference steps MuJoCo)."""
    qp = problems.gen_quadruped_problem(N=N, linearized_friction=linearized_friction)
    rng = np.random.default_rng(seed)
    phases = rng.uniform(0.0, 0.8, 16)
    idx = np.arange(batch) % 16
    dyn = lambda i: tuple(np.stack(a)[idx] for a in zip(*[qp.dynamics(ph + i * qp.dt) for ph in phases]))
    x0 = qp.x_des + rng.standard_normal((batch, 12)) * np.array([.02, .02, .02, .05, .05, .05, .3, .3, .1, .3, .3, .3])
    A, Bm, d = dyn(0)
    sv = api.ALTROSolver(mpc.quadruped_problem(qp, x0, A, Bm, d), api.SolverOptions(**problems.QUADRUPED_OPTS))
    api.solve(sv)
    t, it, ok = [], [], []
    for i in range(1, steps + 1):
        X = api.states(sv)
        xn = X[:, 1] + 1e-3 * rng.standard_normal((batch, 12))
        A, Bm, d = dyn(i)
        api.set_dynamics(sv, api.LinearModel(A, Bm, d, dt=qp.dt, per_knot=True))
        api.set_initial_state(sv, xn)
        api.shift_fill(sv, True, True)
        api.solve(sv)
        st = api.stats(sv)
        t.append(st.tsolve_ms); it.append(st.iterations.copy()); ok.append(st.status == api.SOLVE_SUCCEEDED)
    return _result(t, it, ok, batch)


def summarise(res):
    it = np.asarray(res["iter"])
    return {"batch": int(res["batch"]), "steps": int(it.shape[0]), "iterations_median": float(np.median(it)),
            "iterations_mean": float(it.mean()), "iterations_max": int(it.max()),
            "solve_succeeded_frac": float(np.mean(res["solve_succeeded"])),
            "ms_per_step_median": float(np.median(res["time"])), "us_per_solve_median": float(np.median(res["time_us_per_solve"]))}
