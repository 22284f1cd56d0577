"""Ad-hoc throughput check: B instances, S MPC steps enqueued back-to-back."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
S = int(sys.argv[2]) if len(sys.argv) > 2 else 20
t0 = time.time()
pb = altro.problems.gen_random_linear_batch(B, steps=S)
print("gen", time.time() - t0)
mp = altro.mpc.BatchMPC(pb)
mp.initial_solve()
st = altro.stats(mp.solver)
print("cold ms", st.tsolve_ms, "iters hist", np.bincount(st.iterations))
W = 3
for i in range(W):
    mp.step(i)
t0 = time.time()
for i in range(W, S):
    mp.step_async(i)
mp.synchronize()
dt = time.time() - t0
print("steps", S - W, "wall", dt, "ms/step", dt / (S - W) * 1e3, "solves/s", B * (S - W) / dt)
st = altro.stats(mp.solver)
print("last solve ms", st.tsolve_ms, "iters hist", np.bincount(st.iterations), "status hist", np.bincount(st.status), "outer", np.bincount(st.iterations_outer))

# fused: same steps in one launch on a fresh solver
mp2 = altro.mpc.BatchMPC(pb)
mp2.initial_solve()
for i in range(W):
    mp2.step(i)
altro.timing_reset(mp2.solver)
t0 = time.time()
mp2.run_async(S - W, first=W)
mp2.synchronize()
dt = time.time() - t0
print("FUSED steps", S - W, "wall", dt, "ms/step", dt / (S - W) * 1e3, "solves/s", B * (S - W) / dt)
print("fused == stepwise:", np.array_equal(altro.states(mp.solver), altro.states(mp2.solver)))
wcf = altro.wave_cycles(mp2.solver).astype(float)
print('fused per-wave-step kcycles: total %.0f' % (wcf[:,0].mean()/(S-W)/1e3), 'backward %.0f closed %.0f open %.0f todorov %.0f dual %.0f ls %.0f' % tuple(wcf[:, i].mean()/(S-W)/1e3 for i in range(1, 7)))
print('fused phase shares: backward %.3f closed %.3f open %.3f todorov %.3f dual %.3f ls-sweeps %.3f' % tuple(wcf[:, i].sum() / wcf[:, 0].sum() for i in range(1, 7)))
wc2 = wcf[:, 0]
print("fused wave ticks: min %.0f median %.0f mean %.0f max %.0f ; sum/max/nwaves = %.3f" % (wc2.min(), np.median(wc2), wc2.mean(), wc2.max(), wc2.sum() / wc2.max() / len(wc2)))
wc = altro.wave_cycles(mp.solver).astype(float)[:, 0]
print("wave ticks (s_memtime): min %.0f median %.0f mean %.0f p99 %.0f max %.0f ; sum/max/nwaves = %.3f" % (wc.min(), np.median(wc), wc.mean(), np.percentile(wc, 99), wc.max(), wc.sum() / wc.max() / len(wc)))
nb, nr, ntr = altro.work_counters(mp.solver)
print("backward/solve", nb.sum() / (B * (S)), "rollouts/solve", nr.sum() / (B * S), "interp trials/solve", ntr.sum() / (B * S))
