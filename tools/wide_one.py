"""One wide-kernel MPC run for profiling: python tools/wide_one.py n m N B steps"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
n, m, N, B, S = [int(a) for a in sys.argv[1:6]]
pb = altro.problems.gen_random_linear_batch(B, n=n, m=m, N=N, steps=S + 2, seed=10)
mp = altro.mpc.BatchMPC(pb)
mp.initial_solve()
mp.run_async(S, first=0)
mp.synchronize()
print("done", altro.stats(mp.solver).tsolve_ms)
