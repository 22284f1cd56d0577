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
