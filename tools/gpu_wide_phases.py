"""Cycle breakdown of the wide kernel (needs the -DALTRO_WIDE_STAMPS build via ALTRO_HIP_LIB)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
n, m, N, B = [int(a) for a in sys.argv[1:5]]
pb = altro.problems.gen_random_linear_batch(B, n=n, m=m, N=N, steps=8, seed=10)
mp = altro.mpc.BatchMPC(pb)
mp.initial_solve()
mp.step(0)
altro.timing_reset(mp.solver)
t0 = time.perf_counter(); mp.run_async(4, first=1); mp.synchronize(); dt = time.perf_counter() - t0
tb, tr, tg = altro.work_counters(mp.solver)
ns, ni, nok = altro.solve_counters(mp.solver)
print("n=%d m=%d N=%d B=%d: %.1f ms/step; per instance-solve (Mcycles): backward %.3f (gemm part %.3f) rollouts %.3f ; iterations/solve %.2f" % (
    n, m, N, B, 1e3 * dt / 4, tb.mean() / 4e6, tg.mean() / 4e6, tr.mean() / 4e6, ni.sum() / ns.sum()))
st = altro.stats(mp.solver)
print("   backward segments (Mcycles per instance over 4 steps): expansion %.3f  qv+gemms+rows %.3f  factor+solve %.3f  S update, gains %.3f" % tuple(st.cost_trace[:, 12 + i].mean() / 1e6 for i in range(4)))
print("   whole run %.3f  dual updates %.3f  plant step + shift %.3f  todorov %.3f (Mcycles per instance over 4 steps)" % tuple(st.cost_trace[:, 8 + i].mean() / 1e6 for i in range(4)))
