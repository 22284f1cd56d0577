"""BASELINE config 3 end to end on the GPU: rocket landing (second-order cones), batch 4096.
  1. cold solves (N_track knots, goal + max-thrust / thrust-angle / glideslope cones) from varied
     initial states give every instance its own tracking trajectory   (run_simple_rocket.jl:31-67)
  2. conic MPC over a horizon of N_mpc knots, all steps in one launch (run_simple_rocket.jl:120-135,
     simple_rocket.jl:59-82), with the rocket's plant-noise model
A few instances are checked against the CPU oracle."""
import sys, os, time, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "oracle"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
from altro_mpc_icra2021_amd import problems as P
from helpers import ROCKET_COLD_OPTS, ROCKET_MPC_OPTS, rocket_gpu_problem, rocket_oracle
import oracle_py as O

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Nt = int(sys.argv[2]) if len(sys.argv) > 2 else 301
Nm = int(sys.argv[3]) if len(sys.argv) > 3 else 100
S = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dt = 0.05
rp = P.gen_rocket_problem(N=Nt, tf=(Nt - 1) * dt, Qfk=1e4, Rk=1.0, theta_thrust_max=5.0, theta_glideslope=45.0)
rng = np.random.default_rng(1)
x0 = np.tile(rp.x0, (B, 1)) + rng.standard_normal((B, 6)) * np.array([1, 1, 1, .3, .3, .3]) * 0.5
t0 = time.time()
cold = altro.ALTROSolver(rocket_gpu_problem(altro, rp, x0), altro.SolverOptions(**ROCKET_COLD_OPTS))
altro.solve(cold)
st = altro.stats(cold)
tc = time.time() - t0
print("cold: %d instances, N=%d: %.3f s wall, kernel %.1f ms; iterations mean %.1f max %d; status ok %.4f; outer mean %.1f" % (
    B, Nt, tc, st.tsolve_ms, st.iterations.mean(), st.iterations.max(), (st.status == 1).mean(), st.iterations_outer.mean()))
Xt, Ut = altro.states(cold), altro.controls(cold)
cold.close()
tp = P.gen_rocket_problem(N=Nm, tf=dt * (Nm - 1), include_goal=False, theta_thrust_max=5.0, theta_glideslope=45.0)
tp.Q, tp.R, tp.Qf = np.full(6, 10.0), np.full(3, 0.1), np.full(6, 10.0)
assert S + Nm + 1 <= Nt
noise = rng.standard_normal((S, B, 6))
wts = np.array([1e-3] * 3 + [1e-2] * 3)
grp = np.array([0, 0, 0, 1, 1, 1])
prob = rocket_gpu_problem(altro, tp, Xt[:, 0].copy(), Xt[:, :Nm].copy(), Ut[:, :Nm - 1].copy(), U0=Ut[:, :Nm - 1].copy())
mp = altro.mpc.TrackMPC(prob, altro.SolverOptions(**ROCKET_MPC_OPTS), Xt, Ut, noise, (wts, grp))
mp.initial_solve()
W = 3
for i in range(W):
    mp.step(i)
altro.timing_reset(mp.solver)
t0 = time.time()
mp.run_async(S - W, first=W)
mp.synchronize()
dtm = time.time() - t0
ns, ni, nok = altro.solve_counters(mp.solver)
nb, nr, ntr = altro.work_counters(mp.solver)
print(json.dumps({"workload": "rocket_landing MPC (3 second-order cones) n=6 m=3 N=%d batch=%d" % (Nm, B),
                  "steps": S - W, "solves_per_s": B * (S - W) / dtm, "ms_per_step": 1e3 * dtm / (S - W),
                  "iterations_mean": float(ni.sum() / ns.sum()), "succeeded_frac": float(nok.sum() / ns.sum()),
                  "backward_per_solve": float(nb.sum() / ns.sum()), "rollouts_per_solve": float(nr.sum() / ns.sum())}))
# oracle spot check of the last step for a few instances
X, U = altro.states(mp.solver), altro.controls(mp.solver)
x0g = mp.x0()
for b in range(0, B, max(1, B // 3))[:3]:
    o = rocket_oracle(O, tp, Xt[b, 0], ROCKET_MPC_OPTS, Xt[b, :Nm], Ut[b, :Nm - 1], U0=Ut[b, :Nm - 1])
    o.solve()
    for i in range(S):
        xn = o.plant_step()
        nz = noise[i, b] * np.r_[np.full(3, np.linalg.norm(xn[:3]) * 1e-3), np.full(3, np.linalg.norm(xn[3:]) * 1e-2)]
        o.set_initial_state(xn + nz)
        o.set_reference(Xt[b, i + 1:i + 1 + Nm], Ut[b, i + 1:i + Nm])
        o.shift_fill(True, True)
        so = o.solve()
    print("instance %d after %d MPC steps: |X - oracle| %.2e  |U - oracle| %.2e  iterations %d (oracle %d)" % (
        b, S, np.abs(X[b] - o.states()).max(), np.abs(U[b] - o.controls()).max(), altro.stats(mp.solver).iterations[b], so.iterations))
