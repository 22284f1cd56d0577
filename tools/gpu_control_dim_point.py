"""The reference's control-dimension sweep point with the most controls that one 16 x 16 tile holds: n = 30, m = 15, N = 21
(run_random_linear.jl:142-153) -- `wide_kernel<16, false>`.  Ten fused MPC steps after three warm-up steps at the batch given
(default 8192); prints solves/s, kernel time and the iteration statistics.  python tools/gpu_control_dim_point.py [batch] [m]"""
import sys, os, time, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 15
W, K = 3, 10
pb = altro.problems.gen_random_linear_batch(B, n=30, m=m, N=21, steps=W + 2 * K, seed=7)
mp = altro.mpc.BatchMPC(pb)
mp.initial_solve()
for i in range(W): mp.step(i)
out = []
for w in range(2):
    altro.timing_reset(mp.solver)
    t0 = time.perf_counter()
    mp.run_async(K, first=W + w * K); mp.synchronize()
    dt = time.perf_counter() - t0
    ns, ni, nok = altro.solve_counters(mp.solver)
    out.append({"window": w, "solves_per_s": B * K / dt, "kernel_ms": float(altro.timing_get(mp.solver).sum()), "iterations_mean": float(ni.sum() / ns.sum()),
                "solve_succeeded_frac": float(nok.sum() / ns.sum())})
print(json.dumps({"point": "random_linear_mpc n=30 m=%d N=21" % m, "batch": B, "steps": K, "kernel": ("altro_wide::wide_kernel<0, false>" if m > 16 else "altro_wide::wide_kernel<%d, false, 32>" % (16 if m > 12 else 12 if m > 8 else 8 if m > 4 else 4)), "windows": out}))
