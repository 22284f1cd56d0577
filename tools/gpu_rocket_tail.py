"""Where does the rocket MPC launch (BASELINE configs[2]: N_mpc = 100, batch 4096) spend its time?  Per-instance
iteration counts and per-wave cycles of K fused steps.  Usage: gpu_rocket_tail.py [K=20] [B=4096] [kickout=0]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
P, api, mpcm = altro.problems, altro, altro.mpc
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
kick = int(sys.argv[3]) if len(sys.argv) > 3 else 0
W, Nm, Nt, dt = 5, 100, 301, 0.05
rp = P.gen_rocket_problem(N=Nt, tf=(Nt - 1) * dt, Qfk=1e4, Rk=1.0, theta_thrust_max=5.0, theta_glideslope=45.0)
rng = np.random.default_rng(1)
x0 = np.tile(rp.x0, (B, 1)) + rng.standard_normal((B, 6)) * np.array([1, 1, 1, .3, .3, .3]) * 0.5
cold = api.ALTROSolver(mpcm.constrained_problem(rp, x0), api.SolverOptions(**altro.benchmarks.ROCKET_COLD_OPTS))
api.solve(cold)
Xt, Ut = api.states(cold), api.controls(cold)
cold.close()
tp = P.gen_rocket_problem(N=Nm, tf=dt * (Nm - 1), include_goal=False, theta_thrust_max=5.0, theta_glideslope=45.0)
tp.Q, tp.R, tp.Qf = np.full(6, 10.0), np.full(3, 0.1), np.full(6, 10.0)
prob = mpcm.constrained_problem(tp, Xt[:, 0].copy(), Xt[:, :Nm].copy(), Ut[:, :Nm - 1].copy(), U0=Ut[:, :Nm - 1].copy())
mp = mpcm.TrackMPC(prob, api.SolverOptions(**dict(altro.benchmarks.ROCKET_MPC_OPTS, kickout_max_penalty=kick)), Xt, Ut,
                   rng.standard_normal((W + K, B, 6)), (np.array([1e-3] * 3 + [1e-2] * 3), np.array([0, 0, 0, 1, 1, 1])))
mp.initial_solve()
for i in range(W):
    mp.step(i)
altro.timing_reset(mp.solver)
t0 = time.perf_counter(); mp.run_async(K, first=W); mp.synchronize(); dtm = time.perf_counter() - t0
ns, ni, nok = altro.solve_counters(mp.solver)
nb, nr, ntr = altro.work_counters(mp.solver)
print("rocket N_mpc=100 B=%d K=%d kickout=%d: %.1f ms/step, %.0f solves/s; succeeded %.3f" % (B, K, kick, 1e3 * dtm / K, B * K / dtm, nok.sum() / ns.sum()))
print("  iterations per instance over the launch: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %d; backward passes mean %.0f max %d; rollouts mean %.0f; extra trials mean %.0f max %d" % (
    ni.mean(), np.median(ni), np.percentile(ni, 90), np.percentile(ni, 99), ni.max(), nb.mean(), nb.max(), nr.mean(), ntr.mean(), ntr.max()))
st = altro.stats(mp.solver)
print("  last step: iterations mean %.1f max %d, outer mean %.1f max %d, statuses %s" % (st.iterations.mean(), st.iterations.max(), st.iterations_outer.mean(), st.iterations_outer.max(),
      dict(zip(*np.unique(st.status, return_counts=True)))))
wc = altro.wave_cycles(mp.solver)
if wc.size:
    wc = wc.astype(float)
    print("  wave cycles: mean %.1fM p50 %.1fM p99 %.1fM max %.1fM" % (wc[:, 0].mean() / 1e6, np.median(wc[:, 0]) / 1e6, np.percentile(wc[:, 0], 99) / 1e6, wc[:, 0].max() / 1e6))
    names = ["total", "bw4", "closed", "open", "todorov", "dual", "ls", "nlone", "bwlone", "fosweep", "adjoint", "nbw4", "nfo", "naj", "nrc", "nls"]
    print("  mean wave  :", " ".join("%s %.1fM" % (n, wc[:, i].mean() / 1e6) for i, n in enumerate(names)))
    w = int(np.argmax(wc[:, 0]))
    print("  slowest wave:", " ".join("%s %.1fM" % (n, wc[w, i] / 1e6) for i, n in enumerate(names)), "| row iterations", ni.reshape(-1, 4)[w].tolist(), "backward passes", nb.reshape(-1, 4)[w].tolist(), "trials", ntr.reshape(-1, 4)[w].tolist())
w4 = ni.reshape(-1, 4)
print("  per wave: sum of row iterations mean %.0f max %d; max row mean %.0f" % (w4.sum(1).mean(), w4.sum(1).max(), w4.max(1).mean()))
