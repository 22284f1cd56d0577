"""Quadruped MPC (BASELINE configs[4], batch 2048 = two rounds of blocks on 1024 SIMDs): how much of the launch is the order in
which the blocks are dispatched?  Per-instance cycles of consecutive 10-tick launches (stamps build) and a list-scheduling
simulation on 1024 slots: index order, longest-first by the truth, longest-first by the PREVIOUS launch's iteration counts."""
import sys, os, time, heapq
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
P, api, mpcm = altro.problems, altro, altro.mpc
N, B, S, L = 40, 2048, 10, 4
qp = P.gen_quadruped_problem(N=N)
rng = np.random.default_rng(17)
t0 = rng.uniform(0.0, 0.8, B)
x0 = qp.x_des + rng.standard_normal((B, 12)) * np.array([.02, .02, .02, .05, .05, .05, .3, .3, .1, .3, .3, .3])
T = 1 + S * L + N
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
                   rng.standard_normal((1 + S * L, B, 12)), (np.full(12, 1e-3),))
api.set_dynamics_track(mp.solver, A, Bm, d, step_stride=1)
api.initial_controls(mp.solver, np.tile(qp.u_hover, (B, N - 1, 1)))
mp.initial_solve()
mp.step(0)


def simulate(cyc, order, slots=1024):
    h = [0.0] * slots
    heapq.heapify(h)
    for i in order:
        t = heapq.heappop(h)
        heapq.heappush(h, t + cyc[i])
    return max(h)


prev_it = None
for l in range(L):
    altro.timing_reset(mp.solver)
    t_ = time.perf_counter(); mp.run_async(S, first=1 + l * S); mp.synchronize(); dt = time.perf_counter() - t_
    ns, ni, nok = altro.solve_counters(mp.solver)
    st = altro.stats(mp.solver)
    cyc = st.cost_trace[:, 8].astype(float)
    msg = "launch %d: %.1f ms (%.0f solves/s); iterations mean %.1f max %d; cycles per instance mean %.1fM max %.1fM" % (
        l, 1e3 * dt, B * S / dt, ni.mean(), ni.max(), cyc.mean() / 1e6, cyc.max() / 1e6)
    sim = {"index order": simulate(cyc, range(B)), "longest first (truth)": simulate(cyc, np.argsort(-cyc))}
    if prev_it is not None:
        sim["longest first by the previous launch's iterations"] = simulate(cyc, np.argsort(-prev_it, kind="stable"))
        msg += "; corr(iterations, previous launch) %.2f" % np.corrcoef(ni, prev_it)[0, 1]
    print(msg)
    print("    list scheduling on 1024 slots, Mcycles: " + "; ".join("%s %.1f" % (k, v / 1e6) for k, v in sim.items()) + "  (kernel: %.1f)" % (dt * 2.4e3))
    prev_it = ni.astype(float)
