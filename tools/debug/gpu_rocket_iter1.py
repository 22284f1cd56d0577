"""Localise a GPU/oracle difference: run MPC step 0 of the rocket problem with a cap of K total
iLQR iterations on both sides and print where the trajectories start to differ."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "oracle"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
from altro_mpc_icra2021_amd import problems as P
from helpers import ROCKET_COLD_OPTS, ROCKET_MPC_OPTS, rocket_gpu_problem, rocket_oracle
import oracle_py as O
B, Nt, Nm = 8, 301, int(sys.argv[1]) if len(sys.argv) > 1 else 100
caps = [int(a) for a in sys.argv[2:]] or [0, 1]
import json
EXTRA = json.loads(os.environ.get('ALTRO_DBG_OPTS', '{}'))
dt = 0.05
rp = P.gen_rocket_problem(N=Nt, tf=(Nt - 1) * dt, Qfk=1e4, Rk=1.0, theta_thrust_max=5.0, theta_glideslope=45.0)
rng = np.random.default_rng(1)
x0 = np.tile(rp.x0, (B, 1)) + rng.standard_normal((B, 6)) * np.array([1, 1, 1, .3, .3, .3]) * 0.5
cold = altro.ALTROSolver(rocket_gpu_problem(altro, rp, x0), altro.SolverOptions(**ROCKET_COLD_OPTS))
altro.solve(cold)
Xt, Ut = altro.states(cold), altro.controls(cold)
tp = P.gen_rocket_problem(N=Nm, tf=dt * (Nm - 1), include_goal=False, theta_thrust_max=5.0, theta_glideslope=45.0)
tp.Q, tp.R, tp.Qf = np.full(6, 10.0), np.full(3, 0.1), np.full(6, 10.0)
noise = rng.standard_normal((2, B, 6))
wts = np.array([1e-3] * 3 + [1e-2] * 3); grp = np.array([0, 0, 0, 1, 1, 1])
for cap in caps:
    prob = rocket_gpu_problem(altro, tp, Xt[:, 0].copy(), Xt[:, :Nm].copy(), Ut[:, :Nm - 1].copy(), U0=Ut[:, :Nm - 1].copy())
    mp = altro.mpc.TrackMPC(prob, altro.SolverOptions(**ROCKET_MPC_OPTS), Xt, Ut, noise, (wts, grp))
    mp.initial_solve()
    X0g, U0g = altro.states(mp.solver), altro.controls(mp.solver)
    lam_g = [altro.get_duals(mp.solver, c) for c in range(len(tp.constraints))]
    altro.set_options(mp.solver, iterations=cap, **EXTRA)
    mp.step(0)
    st = altro.stats(mp.solver); Xg, Ug = altro.states(mp.solver), altro.controls(mp.solver); x0g = mp.x0()
    print("==== cap", cap)
    # stepwise host-driven sequence of the same MPC step (set_initial_state / update_trajectory / shift_fill / solve)
    sw = altro.ALTROSolver(rocket_gpu_problem(altro, tp, Xt[:, 0].copy(), Xt[:, :Nm].copy(), Ut[:, :Nm - 1].copy(), U0=Ut[:, :Nm - 1].copy()),
                           altro.SolverOptions(**ROCKET_MPC_OPTS))
    altro.solve(sw)
    altro.set_initial_state(sw, x0g); altro.update_trajectory(sw, Xt[:, 1:1 + Nm].copy(), Ut[:, 1:Nm].copy())
    altro.shift_fill(sw, True, True)
    Upre = altro.controls(sw).copy(); Xpre = np.zeros((B, Nm, 6)); Xpre[:, 0] = x0g
    for kk in range(Nm - 1): Xpre[:, kk + 1] = Xpre[:, kk] @ tp.A.T + Upre[:, kk] @ tp.Bm.T + tp.f
    lam_s = [altro.get_duals(sw, c) for c in range(len(tp.constraints))]; U_s = altro.controls(sw)
    altro.set_options(sw, iterations=cap, **EXTRA); altro.solve(sw)
    Xs, Us = altro.states(sw), altro.controls(sw); sts = altro.stats(sw); Kg, dg = altro.gains(sw)
    for b in range(B):
        o = rocket_oracle(O, tp, Xt[b, 0], ROCKET_MPC_OPTS, Xt[b, :Nm], Ut[b, :Nm - 1], U0=Ut[b, :Nm - 1])
        o.solve()
        e0 = max(np.abs(o.states() - X0g[b]).max(), np.abs(o.controls() - U0g[b]).max())
        el = max(np.abs(np.asarray(o.duals(o.con_ids[c])).ravel() - np.asarray(lam_g[c][b]).ravel()).max() for c in range(len(tp.constraints)))
        xn = o.plant_step()
        nz = noise[0, b] * np.r_[np.full(3, np.linalg.norm(xn[:3]) * 1e-3), np.full(3, np.linalg.norm(xn[3:]) * 1e-2)]
        o.set_initial_state(xn + nz); o.set_reference(Xt[b, 1:1 + Nm], Ut[b, 1:Nm]); o.shift_fill(True, True)
        o.set_opts(O.default_opts(**dict(ROCKET_MPC_OPTS, iterations=cap, **EXTRA)))
        print('   post-shift dual err', [float(np.abs(np.asarray(o.duals(o.con_ids[c])).ravel() - lam_s[c][b].ravel()).max()) for c in range(len(tp.constraints))], 'U err', np.abs(o.controls() - U_s[b]).max())
        so = o.solve()
        Ko, do = o.gains(); eK = np.abs(Kg[b] - Ko).max(axis=(1, 2)); ed = np.abs(dg[b] - do).max(axis=1)
        print('   gains: |K| %.2e |d| %.2e  errK %.2e @%d errd %.2e @%d; errd by knot (last 8):' % (np.abs(Ko).max(), np.abs(do).max(), eK.max(), eK.argmax(), ed.max(), ed.argmax()), ed[-8:], 'errK last 4', eK[-4:])
        bad = np.nonzero(ed > 1e-8 * (1 + np.abs(do).max()))[0]
        if len(bad):
            kb = bad.max()
            print('   last bad knot', kb, 'errd', ed[max(kb-1,0):kb+3], 'errK', eK[max(kb-1,0):kb+3])
            # cone status on the pre-iteration trajectory (the shifted warm start rolled out): use U_s and oracle x0
            for ci, c in enumerate(tp.constraints):
                if c.kind != P.SOC: continue
                for kk in range(max(kb - 1, 0), min(kb + 3, Nm)):
                    if kk < c.k_first or kk > c.k_last: continue
                    A = c.A if c.A.ndim == 2 else c.A[kk - c.k_first]; bb = c.b if c.b.ndim == 1 else c.b[kk - c.k_first]
                    z = np.r_[Xpre[b][kk], Upre[b][min(kk, Nm - 2)]]
                    v = A @ z + bb
                    print('      con', ci, 'knot', kk, 'v', v, '|vbar|-t', np.linalg.norm(v[:-1]) - v[-1], '|vbar|+t', np.linalg.norm(v[:-1]) + v[-1])
        dX = np.abs(Xg[b] - o.states()).max(axis=1); dU = np.abs(Ug[b] - o.controls()).max(axis=1)
        print("   stepwise-vs-oracle dX %.2e dU %.2e J %.10g | fused-vs-stepwise dX %.2e dU %.2e" % (
            np.abs(Xs[b] - o.states()).max(), np.abs(Us[b] - o.controls()).max(), sts.cost[b], np.abs(Xs[b] - Xg[b]).max(), np.abs(Us[b] - Ug[b]).max()))
        print("inst %d: initial-solve err %.1e dual err %.1e | J o/g %.10g %.10g  cmax %.3e %.3e  maxdX %.2e @%d  maxdU %.2e @%d  first dU>1e-9 @%s" % (
            b, e0, el, so.cost, st.cost[b], so.c_max, st.c_max[b], dX.max(), dX.argmax(), dU.max(), dU.argmax(),
            (np.nonzero(dU > 1e-9)[0][:1])))
