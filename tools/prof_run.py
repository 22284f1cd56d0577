"""Small fixed workload for rocprofv3: batch 8192, 3 warm-up steps + one fused launch of 20 steps."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
S = 23
pb = altro.problems.gen_random_linear_batch(8192, steps=S)
mp = altro.mpc.BatchMPC(pb)
mp.initial_solve()
for i in range(3):
    mp.step(i)
mp.run_async(S - 3, first=3)
mp.synchronize()
print("done", altro.stats(mp.solver).tsolve_ms)
