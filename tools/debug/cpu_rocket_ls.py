"""Rocket landing MPC (BASELINE configs[2], N_mpc = 100) on the ORACLE: for every REJECTED line-search trial (J >= J_prev),
after which fraction of the knots would a sweep know it -- partial cost plus a lower bound of the remaining terms (each AL
term >= -|lambda|^2 / 2 mu) already above J_prev?  (what an early exit of the trial sweeps could save)"""
import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
import oracle_py
from helpers import rocket_oracle, ROCKET_COLD_OPTS, ROCKET_MPC_OPTS
P = altro.problems
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 25
Nt, dt, Nm = 301, 0.05, 100
L = oracle_py.lib()
L.orc_debug_ls_trace.restype = C.c_int
L.orc_debug_ls_trace.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
rp = P.gen_rocket_problem(N=Nt, tf=(Nt - 1) * dt, Qfk=1e4, Rk=1.0, theta_thrust_max=5.0, theta_glideslope=45.0)
rng = np.random.default_rng(1)
rows = []      # (iterations of the solve, x_suffix, x_crude, J - J_prev, x_reverse)
for b in range(B):
    x0 = rp.x0 + rng.standard_normal(6) * np.array([1, 1, 1, .3, .3, .3]) * 0.5
    cold = rocket_oracle(oracle_py, rp, x0, ROCKET_COLD_OPTS)
    assert cold.solve().status == 1
    Xt, Ut = cold.states(), cold.controls()
    tp = P.gen_rocket_problem(N=Nm, tf=dt * (Nm - 1), include_goal=False, theta_thrust_max=5.0, theta_glideslope=45.0)
    tp.Q, tp.R, tp.Qf = np.full(6, 10.0), np.full(3, 0.1), np.full(6, 10.0)
    o = rocket_oracle(oracle_py, tp, Xt[0], ROCKET_MPC_OPTS, Xt[:Nm], Ut[:Nm - 1], U0=Ut[:Nm - 1])
    o.solve()
    buf = np.zeros(4 * 20000)
    for i in range(S):
        xn = o.plant_step()
        nz = rng.standard_normal(6) * np.r_[np.full(3, np.linalg.norm(xn[:3]) * 1e-3), np.full(3, np.linalg.norm(xn[3:]) * 1e-2)]
        o.set_initial_state(xn + nz)
        o.set_reference(Xt[i + 1:i + 1 + Nm], Ut[i + 1:i + Nm])
        o.shift_fill(True, True)
        L.orc_debug_ls_trace(o.h, buf.ctypes.data_as(C.POINTER(C.c_double)), 20000)
        so = o.solve()
        n = min(L.orc_debug_ls_trace(o.h, buf.ctypes.data_as(C.POINTER(C.c_double)), 20000), 20000)
        for r in buf[:4 * n].reshape(-1, 4):
            rows.append((so.iterations, r[0], r[1], r[2], r[3]))
    print("instance %d done, %d rejected trials so far" % (b, len(rows)), flush=True)
rows = np.array(rows)
for name, sel in (("all solves", rows[:, 0] >= 0), ("solves of 50+ iterations (the stragglers)", rows[:, 0] >= 50), ("solves of < 20 iterations", rows[:, 0] < 20)):
    r = rows[sel]
    if not len(r):
        continue
    print("%s: %d rejected trials; knots walked before the rejection is certain: suffix bound mean %.2f (never: %.2f), crude bound mean %.2f (never: %.2f); "
          "J - J_prev median %.3g; sweeping from the LAST knot down: mean %.2f (never: %.2f)" % (name, len(r), r[:, 1].mean(), (r[:, 1] >= 1).mean(), r[:, 2].mean(), (r[:, 2] >= 1).mean(),
                                                            np.median(r[:, 3]), r[:, 4].mean(), (r[:, 4] >= 1).mean()))
