"""Find the first MPC step where the GPU conic path and the oracle diverge (rocket, N_mpc=100)."""
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
B, Nt, Nm, S = 8, 301, int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 30
TH = float(os.environ.get('ROCKET_MPC_THETA', '5.0'))
dt = 0.05
rp = P.gen_rocket_problem(N=Nt, tf=(Nt - 1) * dt, Qfk=1e4, Rk=1.0, theta_thrust_max=5.0, theta_glideslope=45.0)
rng = np.random.default_rng(1)
x0 = np.tile(rp.x0, (B, 1)) + rng.standard_normal((B, 6)) * np.array([1, 1, 1, .3, .3, .3]) * 0.5
cold = altro.ALTROSolver(rocket_gpu_problem(altro, rp, x0), altro.SolverOptions(**ROCKET_COLD_OPTS))
altro.solve(cold)
Xt, Ut = altro.states(cold), altro.controls(cold)
tp = P.gen_rocket_problem(N=Nm, tf=dt * (Nm - 1), include_goal=False, theta_thrust_max=TH, theta_glideslope=45.0)
tp.Q, tp.R, tp.Qf = np.full(6, 10.0), np.full(3, 0.1), np.full(6, 10.0)
noise = rng.standard_normal((S, B, 6))
wts = np.array([1e-3] * 3 + [1e-2] * 3); grp = np.array([0, 0, 0, 1, 1, 1])
prob = rocket_gpu_problem(altro, tp, Xt[:, 0].copy(), Xt[:, :Nm].copy(), Ut[:, :Nm - 1].copy(), U0=Ut[:, :Nm - 1].copy())
mp = altro.mpc.TrackMPC(prob, altro.SolverOptions(**ROCKET_MPC_OPTS), Xt, Ut, noise, (wts, grp))
mp.initial_solve()
orcs = [rocket_oracle(O, tp, Xt[b, 0], ROCKET_MPC_OPTS, Xt[b, :Nm], Ut[b, :Nm - 1], U0=Ut[b, :Nm - 1]) for b in range(B)]
for o in orcs: o.solve()
done = set()
for i in range(S):
    mp.step(i)
    st = altro.stats(mp.solver); X = altro.states(mp.solver); x0g = mp.x0(); at = altro.alpha_trace(mp.solver)
    for b in range(B):
        if b in done: continue
        o = orcs[b]
        xn = o.plant_step()
        nz = noise[i, b] * np.r_[np.full(3, np.linalg.norm(xn[:3]) * 1e-3), np.full(3, np.linalg.norm(xn[3:]) * 1e-2)]
        o.set_initial_state(xn + nz); o.set_reference(Xt[b, i + 1:i + 1 + Nm], Ut[b, i + 1:i + Nm]); o.shift_fill(True, True)
        so = o.solve()
        err = np.abs(X[b] - o.states()).max()
        if err > 1e-6 or so.status != 1 or st.status[b] != 1:
            k = min(so.iterations, 16)
            print("step %d inst %d: err %.2e x0err %.1e iters o/g %d %d outer %d %d status %d %d cmax %.2e %.2e" % (
                i, b, err, np.abs(x0g[b] - (xn + nz)).max(), so.iterations, st.iterations[b], so.iterations_outer, st.iterations_outer[b], so.status, st.status[b], so.c_max, st.c_max[b]))
            print("   J o:", np.array(so.J[:k])); print("   J g:", st.cost_trace[b, :k])
            print("   alpha g:", at[b, :k])
            print("   alpha o:", np.array(so.alpha[:k]), " cmax_outer o:", np.array(so.c_max_outer[:so.iterations_outer]))
            done.add(b)
print("instances that diverged or failed:", sorted(done))
