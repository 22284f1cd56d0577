"""Cycle breakdown of the wide kernel on the quadruped MPC loop (needs the -DALTRO_WIDE_STAMPS build via ALTRO_HIP_LIB).
Usage: gpu_wide_quadruped.py [N=40] [B=2048] [steps=4]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
P, api, mpcm = altro.problems, altro, altro.mpc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
S = int(sys.argv[3]) if len(sys.argv) > 3 else 4
qp = P.gen_quadruped_problem(N=N)
rng = np.random.default_rng(17)
t0 = rng.uniform(0.0, 0.8, B)
x0 = qp.x_des + rng.standard_normal((B, 12)) * np.array([.02, .02, .02, .05, .05, .05, .3, .3, .1, .3, .3, .3])
T = 1 + S + N
A, Bm, d = np.zeros((B, T, 12, 12)), np.zeros((B, T, 12, 12)), np.zeros((B, T, 12))
cache = {}
for b in range(B):
    for t in range(T):
        c = tuple(P.trot_contacts(t0[b] + t * qp.dt))
        if c not in cache:
            cache[c] = P.quadruped_linearize(qp.x_des, np.zeros(12), qp.feet, np.array(c), qp.inertia, qp.mass, qp.dt)
        A[b, t], Bm[b, t], d[b, t] = cache[c]
Nt = T + 1
prob = mpcm.quadruped_problem(qp, x0, A[:, :N - 1], Bm[:, :N - 1], d[:, :N - 1])
mp = mpcm.TrackMPC(prob, api.SolverOptions(**P.QUADRUPED_OPTS), np.tile(qp.x_des, (B, Nt, 1)), np.zeros((B, Nt - 1, 12)),
                   rng.standard_normal((1 + S, B, 12)), (np.full(12, 1e-3),))
api.set_dynamics_track(mp.solver, A, Bm, d, step_stride=1)
api.initial_controls(mp.solver, np.tile(qp.u_hover, (B, N - 1, 1)))
mp.initial_solve()
mp.step(0)
altro.timing_reset(mp.solver)
t0_ = time.perf_counter(); mp.run_async(S, first=1); mp.synchronize(); dt = time.perf_counter() - t0_
tb, tr, tg = altro.work_counters(mp.solver)
ns, ni, nok = altro.solve_counters(mp.solver)
print("quadruped N=%d B=%d: %.1f ms/step = %.0f solves/s; per instance-solve (Mcycles): backward %.3f (gemm part %.3f) rollouts %.3f ; iterations/solve %.2f max %d" % (
    N, B, 1e3 * dt / S, B * S / dt, tb.mean() / S / 1e6, tg.mean() / S / 1e6, tr.mean() / S / 1e6, ni.sum() / ns.sum(), ni.max()))
if "stamps" not in os.environ.get("ALTRO_HIP_LIB", ""):
    print("   counters per instance-solve (plain build): backward passes %.2f  rollouts %.2f  extra line-search trials %.2f" % (tb.mean() / S, tr.mean() / S, tg.mean() / S))
print("   costate-confirmed iterations per instance-solve %.3f" % (altro.confirm_counter(mp.solver).mean() / S))
st = altro.stats(mp.solver)
print("   backward segments (Mcycles per instance over %d steps): expansion %.3f  qv+gemms+rows %.3f  factor+solve %.3f  S update, gains %.3f" % ((S,) + tuple(st.cost_trace[:, 12 + i].mean() / 1e6 for i in range(4))))
print("   whole run %.3f  dual updates %.3f  plant step + shift %.3f  todorov %.3f (Mcycles per instance over %d steps)" % (tuple(st.cost_trace[:, 8 + i].mean() / 1e6 for i in range(4)) + (S,)))
it = ni.astype(float)
print("   per-instance iterations over the launch: mean %.1f p99 %.0f max %d" % (it.mean(), np.percentile(it, 99), it.max()))
