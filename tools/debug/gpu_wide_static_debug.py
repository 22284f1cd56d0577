"""Compare the wide kernel with and without the LDS-resident constraint table on the rocket MPC test scenario."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "oracle"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
import numpy as np
os.environ["ALTRO_FORCE_WIDE"] = "1"
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
from altro_mpc_icra2021_amd import problems as P
from helpers import ROCKET_COLD_OPTS, ROCKET_MPC_OPTS, rocket_gpu_problem, rocket_oracle
import oracle_py as O
B, Nm, S = 5, 21, 4
rp = P.gen_rocket_problem(N=61, tf=15.0, Qfk=1e4, Rk=1.0, theta_thrust_max=5.0, theta_glideslope=45.0)
cold = rocket_oracle(O, rp, rp.x0, ROCKET_COLD_OPTS); cold.solve()
Xt, Ut = cold.states(), cold.controls()
tp = P.gen_rocket_problem(N=Nm, tf=rp.dt * (Nm - 1), include_goal=False, theta_thrust_max=5.0, theta_glideslope=45.0)
tp.Q, tp.R, tp.Qf = np.full(6, 10.0), np.full(3, 0.1), np.full(6, 10.0)
x0 = np.tile(Xt[0], (B, 1)); Xr = np.tile(Xt[:Nm], (B, 1, 1)); Ur = np.tile(Ut[:Nm - 1], (B, 1, 1))
res = {}
for mode in ("0", "1"):
    os.environ["ALTRO_WIDE_STATIC_MASK"] = sys.argv[1] if mode == "0" else "0"
    rng = np.random.default_rng(3)
    sv = altro.ALTROSolver(rocket_gpu_problem(altro, tp, x0, Xr, Ur, U0=Ur.copy()), altro.SolverOptions(**ROCKET_MPC_OPTS))
    altro.solve(sv)
    orcs = [rocket_oracle(O, tp, x0[b], ROCKET_MPC_OPTS, Xr[b], Ur[b], U0=Ur[b]) for b in range(B)]
    for o in orcs: o.solve()
    out = []
    for i in range(S):
        x0n = np.zeros((B, 6))
        for b in range(B):
            xn = orcs[b].plant_step()
            x0n[b] = xn + np.r_[rng.standard_normal(3) * np.linalg.norm(xn[:3]) / 1000.0, rng.standard_normal(3) * np.linalg.norm(xn[3:]) / 100.0]
            orcs[b].set_initial_state(x0n[b]); orcs[b].set_reference(Xt[i + 1:i + 1 + Nm], Ut[i + 1:i + Nm]); orcs[b].shift_fill(True, True)
        altro.set_initial_state(sv, x0n)
        altro.update_trajectory(sv, np.tile(Xt[i + 1:i + 1 + Nm], (B, 1, 1)), np.tile(Ut[i + 1:i + Nm], (B, 1, 1)))
        altro.shift_fill(sv, True, True)
        lam = [altro.get_duals(sv, c).copy() for c in range(3)]
        altro.solve(sv)
        st = altro.stats(sv)
        sos = [o.solve() for o in orcs]
        Xg, Ug = altro.states(sv), altro.controls(sv); lam2 = [altro.get_duals(sv, c).copy() for c in range(3)]
        out.append((st.iterations.copy(), np.array([s.iterations for s in sos]), st.cost_trace[:, :3].copy(), np.array([s.J[:3] for s in sos]), lam, Xg, Ug, lam2, st.cost_trace.copy(), [o.states() for o in orcs]))
    res[mode] = out
for i in range(S):
    a, b = res["0"][i], res["1"][i]
    print("step", i, "iters static", a[0], "nostatic", b[0], "oracle", a[1])
    print("   J0 static", a[2][:, 0], "\n   J0 nostat", b[2][:, 0], "\n   J0 oracle", a[3][:, 0])
    print("   pre-solve dual diff static-vs-nostatic:", [float(np.abs(x - y).max()) for x, y in zip(a[4], b[4])])
    dX = np.abs(a[5] - b[5]).max(axis=(1, 2)); dU = np.abs(a[6] - b[6]).max(axis=(1, 2))
    print("   post-solve |dX| per instance", dX, " |dU|", dU)
    print("   post-solve dual diff per constraint, per knot (instance 1):", [np.abs(x[1] - y[1]).max(axis=-1).round(12).tolist() for x, y in zip(a[7], b[7])][1])
    k = min(a[0][1], 8)
    print("   J trace inst 1 static", a[8][1, :k], "\n                nostat", b[8][1, :k])
    print("   static vs oracle |dX| inst1 %.2e ; nostatic vs oracle %.2e" % (np.abs(a[5][1] - a[9][1]).max(), np.abs(b[5][1] - b[9][1]).max()))
    if i >= 1: break
