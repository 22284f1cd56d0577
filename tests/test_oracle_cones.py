"""CPU pins of the oracle's conic path (LINEAR equality + second-order cones) on the reference's
rocket-landing problem (benchmarks/rocket_landing/rocket_landing_problem.jl:44-186,
run_simple_rocket.jl:31-62,112-116)."""
import numpy as np
from altro_mpc_icra2021_amd import problems as P
from helpers import ROCKET_COLD_OPTS, admm_conic_qp, rocket_oracle


def test_rocket_cold_solve_is_feasible_and_cones_are_active(oracle):
    """The reference's own post-solve checks (run_simple_rocket.jl:112-116): max thrust below the
    bound, thrust angle below 5 deg, glideslope below 45 deg -- and the goal is reached."""
    rp = P.gen_rocket_problem(N=61, tf=15.0, Qfk=1e4, Rk=1.0, theta_thrust_max=5.0, theta_glideslope=45.0)
    s = rocket_oracle(oracle, rp, rp.x0, ROCKET_COLD_OPTS)
    st = s.solve()
    assert st.status == 1 and st.c_max < 1e-5
    assert 3 <= st.iterations_outer <= 10 and st.iterations < 60
    X, U = s.states(), s.controls()
    assert np.abs(X[-1]).max() < 1e-5
    assert np.linalg.norm(U, axis=1).max() <= 196.2 * (1 + 1e-6)
    ang = np.degrees(np.arctan2(np.linalg.norm(U[:, :2], axis=1), U[:, 2]))
    assert ang.max() <= 5.0 + 1e-4 and ang.max() > 4.99          # active
    gl = np.degrees(np.arctan2(np.linalg.norm(X[7:-1, :2], axis=1), X[7:-1, 2]))
    assert gl.max() <= 45.0 + 1e-3
    for k in range(rp.N - 1):
        assert np.allclose(X[k + 1], rp.A @ X[k] + rp.Bm @ U[k] + rp.f, atol=1e-10)


def test_small_conic_problem_matches_independent_solver(oracle):
    """Converged AL-iLQR solution == ADMM solution of the same SOCP condensed in U (the
    reference's validation method: ALTRO vs COSMO / ECOS, simple_rocket.jl:183-203)."""
    rp = P.gen_rocket_problem(N=16, tf=15.0, glide_recover_k=3)      # RocketProblem defaults otherwise
    opts = dict(ROCKET_COLD_OPTS, constraint_tolerance=1e-8, cost_tolerance=1e-10, cost_tolerance_intermediate=1e-8,
                gradient_tolerance=1e-6, gradient_tolerance_intermediate=1e-6, penalty_scaling=50.0)
    s = rocket_oracle(oracle, rp, rp.x0, opts)
    st = s.solve()
    assert st.status == 1
    N, n, m = rp.N, rp.n, rp.m
    nu = (N - 1) * m
    # X = x_free + Gam U
    Gam = np.zeros((N * n, nu))
    xfree = np.zeros((N, n))
    xfree[0] = rp.x0
    for k in range(1, N):
        xfree[k] = rp.A @ xfree[k - 1] + rp.f
        Gam[k * n:(k + 1) * n] = rp.A @ Gam[(k - 1) * n:k * n]
        Gam[k * n:(k + 1) * n, (k - 1) * m:k * m] += rp.Bm
    wx = np.concatenate([np.full((N - 1) * n, rp.dt) * np.tile(rp.Q, N - 1), rp.Qf])
    Pm = Gam.T @ (wx[:, None] * Gam) + rp.dt * np.diag(np.tile(rp.R, N - 1))
    q = Gam.T @ (wx * xfree.reshape(-1))
    rows, hs, cones = [], [], []
    for c in rp.constraints:
        for k in range(c.k_first, c.k_last + 1):
            Ax, Au = c.A[:, :n], c.A[:, n:]
            Gk = Ax @ Gam[k * n:(k + 1) * n]
            if k < N - 1:
                Gk = Gk.copy()
                Gk[:, k * m:(k + 1) * m] += Au
            rows.append(Gk)
            hs.append(Ax @ xfree[k] + c.b)
            cones.append(("soc" if c.kind == P.SOC else "zero", c.A.shape[0]))
    G, h = np.vstack(rows), np.concatenate(hs)
    Ua, it = admm_conic_qp(Pm, q, G, h, cones, rho=0.1)
    Uo = s.controls().reshape(-1)
    f = lambda U: 0.5 * U @ Pm @ U + q @ U
    assert abs(f(Ua) - f(Uo)) <= 1e-6 * max(1.0, abs(f(Uo)))
    assert np.abs(Ua - Uo).max() <= 1e-4 * max(1.0, np.abs(Uo).max())


def test_soc_gauss_newton_and_curvature_variants_agree(oracle):
    """The projection-curvature term changes the iterate path, not the converged answer."""
    rp = P.gen_rocket_problem(N=31, tf=15.0, Qfk=1e4, Rk=1.0, theta_thrust_max=5.0, theta_glideslope=45.0)
    sols = []
    for so in (1, 0):
        s = rocket_oracle(oracle, rp, rp.x0, dict(ROCKET_COLD_OPTS, soc_second_order=so))
        st = s.solve()
        assert st.status == 1
        sols.append((s.states(), s.controls(), st.iterations))
    assert np.abs(sols[0][0] - sols[1][0]).max() < 1e-3
    assert np.abs(sols[0][1] - sols[1][1]).max() < 5e-2


def load_grasp_fixture():
    import json
    import os
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "grasp_ref_traj.json")))
    a = [np.array(x["values"]) for x in d["arrays"]]
    y, z = a[0], a[1]
    F1 = np.array(a[2:32])      # 30 x (Fy, Fz)
    F2 = np.array(a[32:62])
    theta = a[62]
    p1 = np.array(a[63:94])     # o_p[1][t], 31 x 3
    return y, z, F1, F2, theta, p1


def test_grasp_cold_solve_matches_reference_trajectory(oracle):
    """Known answer stored by the reference: benchmarks/grasp_optimization/grasp_ref_traj.jld2,
    written by old/altro_cold_solve.jl:102-117 from a cold ALTRO solve of GraspProblem with
    N=31, tf=3 (conic AL + per-knot-varying linear and second-order-cone constraints).  Two pins: the
    optimum (strictly convex problem: at constraint_tolerance 1e-7 the oracle lands within 3e-7 of the
    stored trajectory) and, below, the reference's own solve reproduced to rounding."""
    y, z, F1, F2, theta, p1 = load_grasp_fixture()
    gp = P.gen_grasp_problem(N=31, tf=3.0)
    # the problem restatement itself is pinned by the stored orientation and contact-point data
    assert np.abs(gp.theta - theta).max() < 1e-12
    assert np.abs(np.array(gp.p[0]) - p1).max() < 1e-12
    opts = dict(cost_tolerance=1e-8, cost_tolerance_intermediate=1e-7, constraint_tolerance=1e-7, penalty_initial=1.0,
                penalty_scaling=10.0, iterations=5000, iterations_outer=60, iterations_inner=300,
                gradient_tolerance=1e-5, gradient_tolerance_intermediate=1e-5)
    s = rocket_oracle(oracle, gp, gp.x0, opts)
    st = s.solve()
    assert st.status == 1, (st.status, st.iterations, st.c_max)
    X, U = s.states(), s.controls()
    assert np.abs(X[:, 1] - y).max() < 1e-6
    assert np.abs(X[:, 2] - z).max() < 1e-6
    assert np.abs(U[:, 1:3] - F1).max() < 1e-6
    assert np.abs(U[:, 4:6] - F2).max() < 1e-6
    # THE REFERENCE'S OWN SOLVE (old/altro_cold_solve.jl:79-86): projected_newton is left at its default (true)
    # with projected_newton_tolerance = 1e-5, so Altro runs the AL solver to a constraint tolerance of 1e-5 (the
    # polish tolerance replaces constraint_tolerance for the AL stage) and then skips the polish, because the
    # violation is already below constraint_tolerance = 1e-4.  The stored file is therefore the output of the
    # AL-iLQR path alone at tolerance 1e-5, penalty 1 x 10, cost_tolerance_intermediate 1e-5 -- and the
    # restatement reproduces it TO ROUNDING (1e-14) after 17 iterations in 5 outer iterations: the same iterate
    # path, not just the same optimum.  Both cone-Hessian variants walk it.
    for so2 in (1, 0):
        s2 = rocket_oracle(oracle, gp, gp.x0, dict(cost_tolerance_intermediate=1e-5, constraint_tolerance=1e-5,
                                                   penalty_initial=1.0, penalty_scaling=10.0, soc_second_order=so2))
        st2 = s2.solve()
        assert st2.status == 1 and st2.iterations == 17 and st2.iterations_outer == 5, (st2.iterations, st2.iterations_outer)
        X2, U2 = s2.states(), s2.controls()
        assert np.abs(X2[:, 1] - y).max() < 1e-12 and np.abs(X2[:, 2] - z).max() < 1e-12
        assert np.abs(U2[:, 1:3] - F1).max() < 1e-12 and np.abs(U2[:, 4:6] - F2).max() < 1e-12
    # at the script's nominal constraint_tolerance (1e-4) the AL path stops one outer iteration earlier, 2e-4 away
    s3 = rocket_oracle(oracle, gp, gp.x0, dict(cost_tolerance_intermediate=1e-5, constraint_tolerance=1e-4,
                                               penalty_initial=1.0, penalty_scaling=10.0))
    st3 = s3.solve()
    assert st3.status == 1 and st3.iterations == 15 and st3.iterations_outer == 4
    assert np.abs(s3.states()[:, 1] - y).max() < 1e-3 and np.abs(s3.controls()[:, 1:3] - F1).max() < 1e-3
    assert np.abs(X[:, 0]).max() < 1e-6 and np.abs(U[:, [0, 3]]).max() < 1e-6     # motion stays in the y-z plane


def _grasp_mpc_iterations(oracle, gp, Xt, Ut, Nm, steps, seed, soc_second_order=1):
    """run_grasp_mpc (grasp_mpc.jl:8-104): solve, then per step mpc_update! (1 % plant noise, retarget,
    primal shift, REWRITE of the per-knot constraint data), dual shift, ONE solve! -- no
    benchmark_solve! here, and reset_duals stays true (grasp_benchmark.jl:26-34)."""
    import copy
    mpc_opts = dict(cost_tolerance=1e-4, cost_tolerance_intermediate=1e-3, constraint_tolerance=1e-4,
                    penalty_initial=10000.0, penalty_scaling=100.0, soc_second_order=soc_second_order)

    def window(k0):
        return [P.ConstraintSpec(c.kind, c.sense, 0, Nm - 2, A=c.A[k0:k0 + Nm - 1].copy(), b=c.b[k0:k0 + Nm - 1].copy())
                for c in gp.constraints[1:]]

    tp = copy.copy(gp)
    tp.N, tp.Q, tp.R, tp.Qf = Nm, np.full(6, 1e3), np.full(6, 1.0), np.full(6, 10.0)   # grasp_benchmark.jl:79-80
    tp.constraints = window(0)
    o = rocket_oracle(oracle, tp, Xt[0], mpc_opts, Xt[:Nm], Ut[:Nm - 1], U0=Ut[:Nm - 1])
    assert o.solve().status == 1
    rng = np.random.default_rng(seed)
    its, ok = [], 0
    for i in range(1, steps + 1):
        xn = o.plant_step()
        o.set_initial_state(xn + rng.standard_normal(6) * np.abs(xn).max() / 100.0)
        o.set_reference(Xt[i:i + Nm], Ut[i:i + Nm - 1])
        o.shift_fill(True, False)
        for ci, c in enumerate(window(i)):
            o.update_constraint_data(o.con_ids[ci], c.A, c.b)
        o.shift_fill(False, True)
        so = o.solve()
        its.append(so.iterations)
        ok += so.status == 1
    return np.array(its), ok


def test_grasp_mpc_iteration_statistics_against_reference(oracle):
    """benchmarks/grasp_optimization/grasp_benchmark_data.jld2 (tests/golden/ref_grasp_mpc_stats.json) holds
    ALTRO's per-step iteration counts of the conic grasp MPC loop for N_mpc = 11..51, three times over (once
    per comparison solver, different noise each time): 15 runs of 200-240 steps, every one with median 3,
    minimum 2, mean 3.3-4.0, maximum 8-20.  The same loop on the restatement (cold solve at N = 251, tracking
    problem, per-step constraint rewrites, duals reset every solve; own noise samples):

      * what is reproduced: every solve SOLVE_SUCCEEDED, minimum 2, the bulk at 2-4 iterations, maxima of the
        same order;
      * OPEN FIDELITY GAP (DESIGN.md "Oracle and parity"): the restatement needs about one iteration more per
        solve on average (mean 4.5 with the Gauss-Newton cone Hessian, 5.5 with the projection-curvature term,
        against 3.3-4.0) -- 20 % of its solves take 7+ iterations where 5 % of the reference's do.  The
        converged answers agree (the stored grasp trajectory is reproduced to 3e-7 above); what differs is
        Altro.jl's iterate path on cold duals at penalty 1e4, which nothing in the reference records.
    Round 4: the AL active-set tolerance (Altro's active_set_tolerance_al = 1e-3: rows with c >= -tol count as active)
    was tried in the restatement and moves the means by less than 0.2 either way -- not the cause.  The Gauss-Newton
    variant is the closer one HERE (4.3-4.9) but cannot be the default: on BASELINE configs[2] (rocket landing,
    N_mpc = 100, batch 4096) it triples the iterations per solve (102 against 34) and loses 2 % of the solves
    (tools/debug/gpu_rocket_so2.py), while the curvature term reproduces the stored rocket step.
    The bounds below pin today's behaviour so that a change of the restatement shows up here."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_grasp_mpc_stats.json")))
    runs = gold["runs"]
    assert len(runs) == 15 and all(r["iter_median"] == 3.0 and r["iter_min"] == 2 for r in runs)
    mean_lo, mean_hi = min(r["iter_mean"] for r in runs), max(r["iter_mean"] for r in runs)
    assert 3.3 < mean_lo and mean_hi < 4.05
    max_hi = max(r["iter_max"] for r in runs)
    gp = P.gen_grasp_problem(N=251, tf=6.0)                     # GraspProblem(o, 251): grasp_benchmark.jl:72, grasp_problem.jl:1
    cold = rocket_oracle(oracle, gp, gp.x0, dict(cost_tolerance=1e-6, cost_tolerance_intermediate=1e-4, constraint_tolerance=1e-6,
                                                 iterations=5000, iterations_outer=60, iterations_inner=300))
    assert cold.solve().status == 1
    Xt, Ut = cold.states(), cold.controls()
    for Nm, seed in ((11, 1), (31, 2), (51, 3)):
        steps = 251 - Nm
        for so2, gap, med_hi in ((0, 1.0, 4), (1, 1.9, 5)):     # Gauss-Newton cone Hessian / with the curvature term (observed: 4.3-4.9 / 5.1-5.7)
            its, ok = _grasp_mpc_iterations(oracle, gp, Xt, Ut, Nm, steps, seed, so2)
            assert ok >= steps - 2, (Nm, ok)                    # SOLVE_SUCCEEDED (the reference's loop does not check)
            assert its.min() == 2 and 3 <= np.median(its) <= med_hi, (Nm, so2, np.median(its), its.min())
            assert mean_lo <= its.mean() <= mean_hi + gap, (Nm, so2, its.mean(), mean_lo, mean_hi)
            assert (its <= 4).mean() >= 0.4 and its.max() <= 3 * max_hi, (Nm, so2, its.max())


def test_rocket_mpc_step_error_vs_solver_tolerance_follows_the_reference_table(oracle):
    """benchmarks/rocket_landing/rocket.jld2 + figures/rocket_solver_tol.tikz (tests/golden/ref_rocket_step.json):
    run_simple_rocket.jl:146-206 solves ONE conic MPC step (N_mpc = 21, first solve at tol0 = 1e-6, then
    mpc_update with the position / velocity noise of simple_rocket.jl:65-71) at solver tolerances 1e-2 .. 1e-12
    (cost, constraint and both gradient tolerances all set to tol) and stores the trajectory error against a
    tight solution: 0.47 at 1e-2, then a plateau of 5.4e-7 from 1e-4 on -- ALTRO's conic AL is at its answer as
    soon as the duals are warm -- and the stored `res` of one such step (9 iterations, ALTRO-vs-conic-solver
    error 1.5e-9 in the states, 4e-7 in the controls).  The Julia noise sample is not reproducible, so the
    table is pinned in shape: coarse at 1e-2, at the 1e-6 level from tol = 1e-6 on, a handful of iterations."""
    import json
    import os
    from helpers import ROCKET_COLD_OPTS, ROCKET_MPC_OPTS
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_rocket_step.json")))
    tab = dict((t, e) for t, e in gold["tol_comp"]["ALTRO"])
    assert gold["iter_altro_other"][0] == 9 and tab[1e-2] > 0.1 and all(tab[t] < 1e-6 for t in (1e-4, 1e-6, 1e-8, 1e-10, 1e-12))
    assert gold["err_traj_state_control_dynamics"][0] < 1e-8 and gold["err_traj_state_control_dynamics"][1] < 1e-6
    Nt, dt, Nm = 301, 0.05, 21
    rp = P.gen_rocket_problem(N=Nt, tf=(Nt - 1) * dt, Qfk=1e4, Rk=1.0, theta_thrust_max=5.0, theta_glideslope=45.0)
    cold = rocket_oracle(oracle, rp, rp.x0, ROCKET_COLD_OPTS)
    assert cold.solve().status == 1
    Xt, Ut = cold.states(), cold.controls()
    tp = P.gen_rocket_problem(N=Nm, tf=dt * (Nm - 1), include_goal=False, theta_thrust_max=5.0, theta_glideslope=45.0)
    tp.Q, tp.R, tp.Qf = np.full(6, 10.0), np.full(3, 0.1), np.full(6, 10.0)           # gen_tracking_problem, mpc.jl:12-14
    keys = ("constraint_tolerance", "cost_tolerance", "cost_tolerance_intermediate", "gradient_tolerance")

    def one_step(tol, seed):
        o0 = dict(ROCKET_MPC_OPTS, **{k: 1e-6 for k in keys})                           # tol0 (simple_rocket.jl:122-128)
        o = rocket_oracle(oracle, tp, Xt[0], o0, Xt[:Nm], Ut[:Nm - 1], U0=Ut[:Nm - 1])
        assert o.solve().status == 1
        o1 = dict(ROCKET_MPC_OPTS, iterations=2000, iterations_outer=60, gradient_tolerance_intermediate=tol, **{k: tol for k in keys})
        o.set_opts(oracle.default_opts(**o1))
        rng = np.random.default_rng(seed)
        xn = o.plant_step()
        o.set_initial_state(xn + np.r_[rng.standard_normal(3) * np.linalg.norm(xn[:3]) / 1000.0,
                                       rng.standard_normal(3) * np.linalg.norm(xn[3:]) / 100.0])
        o.set_reference(Xt[1:1 + Nm], Ut[1:Nm])
        o.shift_fill(True, True)
        so = o.solve()
        return o.states(), o.controls(), so

    for seed in (1, 2):
        Xr, Ur, _ = one_step(1e-13, seed)
        err, its = {}, {}
        for tol in (1e-2, 1e-4, 1e-6, 1e-8, 1e-10):
            X, U, so = one_step(tol, seed)
            assert so.status == 1, (tol, so.status)
            err[tol], its[tol] = max(np.abs(X - Xr).max(), np.abs(U - Ur).max()), so.iterations
        assert err[1e-2] > 1e-3 and err[1e-2] > 50 * err[1e-4], (seed, err)            # coarse, then a sharp drop
        assert all(err[t] < 2e-6 for t in (1e-6, 1e-8, 1e-10)), (seed, err)             # the plateau level of the table
        assert 3 <= its[1e-4] <= 12 and its[1e-10] <= 25, (seed, its)                   # stored step: 9 iterations


def test_projected_newton_polish_on_the_stored_grasp_problem(oracle):
    """the projected-Newton polish (Altro.jl default, off in every live script of the reference; PARITY
    UNPINNED -- the reference stores no trajectory a polish produced).  The script that wrote grasp_ref_traj.jld2
    (old/altro_cold_solve.jl:79-86) leaves it on with projected_newton_tolerance = 1e-5 and constraint_tolerance = 1e-4,
    and the polish is then skipped (the AL stage already ends below 1e-4; reproduced to 1e-12 above).  Forcing it to run --
    AL stage to a LOOSE 1e-2, polish to 1e-6 -- must land on the same optimum: closer to the stored trajectory than the AL
    path alone at that loose tolerance, with the constraints (cones, per-knot equalities and inequalities, goal) and the
    dynamics satisfied to the polish tolerance."""
    y, z, F1, F2, theta, p1 = load_grasp_fixture()
    gp = P.gen_grasp_problem(N=31, tf=3.0)
    base = dict(cost_tolerance_intermediate=1e-5, penalty_initial=1.0, penalty_scaling=10.0)
    loose = rocket_oracle(oracle, gp, gp.x0, dict(base, constraint_tolerance=1e-2))
    sl = loose.solve()
    pol = rocket_oracle(oracle, gp, gp.x0, dict(base, constraint_tolerance=1e-6, projected_newton=1, projected_newton_tolerance=1e-2))
    sp = pol.solve()
    assert sl.status == 1 and sp.status == 1 and sp.pn_ran == 1 and sp.pn_failed == 0
    assert sp.iterations == sl.iterations and sp.iterations_outer == sl.iterations_outer      # same AL stage
    assert sl.c_max > 1e-4 and sp.c_max < 1e-6 and sp.pn_residual < 1e-6
    Xl, Ul, Xp, Up = loose.states(), loose.controls(), pol.states(), pol.controls()

    def dist(X, U):
        return max(np.abs(X[:, 1] - y).max(), np.abs(X[:, 2] - z).max()), max(np.abs(U[:, 1:3] - F1).max(), np.abs(U[:, 4:6] - F2).max())
    (dxl, dul), (dxp, dup) = dist(Xl, Ul), dist(Xp, Up)
    assert dxp < 0.5 * dxl and dup < 0.5 * dul, (dxl, dul, dxp, dup)
    # dynamics: the polish moves states and controls together; the defects stay below its tolerance
    Xn = Xp[:-1] @ gp.A.T + Up @ gp.Bm.T + gp.f
    assert np.abs(Xn - Xp[1:]).max() < 1e-6 and np.abs(Xp[0] - gp.x0).max() < 1e-6
    # with the script's nominal options the polish does not run and the AL path's output is untouched
    nom = rocket_oracle(oracle, gp, gp.x0, dict(base, constraint_tolerance=1e-4, projected_newton=1, projected_newton_tolerance=1e-5))
    sn = nom.solve()
    assert sn.pn_ran == 0 and sn.iterations == 17 and np.abs(nom.states()[:, 1] - y).max() < 1e-12


def test_projected_newton_polish_box_constrained_lq(oracle):
    """Box-constrained LQ tracking problem with saturating controls: AL stage to 1e-3, polish to 1e-8 -- the polished
    trajectory is the tight AL solution (1e-7), bounds and dynamics to 1e-8."""
    import altro_mpc_icra2021_amd as altro
    from helpers import REF_OPTS, make_oracle
    pb = altro.problems.gen_random_linear_batch(2, steps=1, seed=81)
    x0 = pb.window(0)[0][0, 0] + 25.0
    tight = make_oracle(oracle, pb, 0, opts=dict(REF_OPTS, constraint_tolerance=1e-10, cost_tolerance=1e-10, cost_tolerance_intermediate=1e-10, iterations_outer=60))
    tight.set_initial_state(x0)
    assert tight.solve().status == 1
    pol = make_oracle(oracle, pb, 0, opts=dict(REF_OPTS, constraint_tolerance=1e-8, projected_newton=1))
    pol.set_initial_state(x0)
    sp = pol.solve()
    assert sp.status == 1 and sp.pn_ran == 1 and sp.c_max < 1e-8
    X, U = pol.states(), pol.controls()
    assert np.abs(U).max() <= 3.0 + 1e-8
    assert np.abs(X - tight.states()).max() < 1e-7 and np.abs(U - tight.controls()).max() < 1e-7
    assert np.abs(X[:-1] @ pb.A[0].T + U @ pb.Bm[0].T - X[1:]).max() < 1e-8


def test_multiplier_projection_after_the_polish(oracle):
    """SURVEY 8 f4, second half: Altro's multiplier projection after the primal polish -- the least-squares multipliers of
    the polish's active rows D (initial condition, active bounds, dynamics) at the polished trajectory,
    lam = -(D D')^-1 D g, and the stationarity residual ||g + D' lam||_2 before (AL duals, zero elsewhere) and after.
    Checked against a dense numpy restatement: the residual after is ||(I - D'(D D')^-1 D) g||, and at the (polished)
    optimum of the box-constrained LQ problem it vanishes -- the KKT conditions hold with those multipliers."""
    import altro_mpc_icra2021_amd as altro
    from helpers import REF_OPTS, make_oracle
    pb = altro.problems.gen_random_linear_batch(2, n=4, m=2, N=12, steps=1, seed=83)
    x0 = pb.window(0)[0][0, 0] + 12.0
    pol = make_oracle(oracle, pb, 0, opts=dict(REF_OPTS, constraint_tolerance=1e-9, projected_newton=1))
    pol.set_initial_state(x0)
    sp = pol.solve()
    assert sp.status == 1 and sp.pn_ran == 1 and sp.pn_failed == 0 and sp.pn_dual_failed == 0
    X, U = pol.states(), pol.controls()
    n, m, N = 4, 2, 12
    A, Bm, dt = pb.A[0], pb.Bm[0], pb.dt
    Xr, Ur = pb.window(0)[0][0], pb.window(0)[1][0]
    nz = n + m
    nv = N * nz - m                       # z = (x_0, u_0, ..., x_{N-1})
    g = np.zeros(nv)
    for k in range(N):
        g[k * nz:k * nz + n] = (10.0 * dt if k < N - 1 else 10.0) * (X[k] - Xr[k])
        if k < N - 1:
            g[k * nz + n:(k + 1) * nz] = 0.1 * dt * (U[k] - Ur[k])
    rows = []
    for i in range(n):                    # x_0 - x0
        r = np.zeros(nv); r[i] = 1.0; rows.append(r)
    nact = 0
    for k in range(N - 1):
        for side, sgn in ((0, 1.0), (1, -1.0)):          # u - 3 <= 0, -3 - u <= 0, active within 1e-3
            for i in range(m):
                c = sgn * U[k, i] - pb.u_bnd
                if c >= -1e-3:
                    r = np.zeros(nv); r[k * nz + n + i] = sgn; rows.append(r); nact += 1
        for i in range(n):                # A x_k + B u_k - x_{k+1}
            r = np.zeros(nv)
            r[k * nz:k * nz + n] = A[i]; r[k * nz + n:(k + 1) * nz] = Bm[i]; r[(k + 1) * nz + i] -= 1.0
            rows.append(r)
    D = np.array(rows)
    assert nact >= 2                      # the problem saturates: the projection has bound rows to deal with
    lam = -np.linalg.solve(D @ D.T, D @ g)
    res = np.linalg.norm(g + D.T @ lam)
    assert abs(sp.pn_dual_residual - res) <= 1e-7 * max(1.0, np.linalg.norm(g)), (sp.pn_dual_residual, res)
    assert sp.pn_dual_residual < 1e-6 * np.linalg.norm(g) and sp.pn_dual_residual0 > 0.1 * np.linalg.norm(g)
