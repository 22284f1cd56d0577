"""Kernel time of twelve consecutive 20-step windows of the headline closed loop for several builds of the library
(ALTRO_HIP_LIB), each launch right after 200 steps of a scratch copy of the batch (clocks up); separate processes, twice over.
Usage: gpu_lib_windows.py tag=lib.so [tag=lib.so ...]"""
import sys, os, subprocess
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
if len(sys.argv) == 2 and "=" not in sys.argv[1]:
    import numpy as np
    import altro_amd_loader
    import altro_mpc_icra2021_amd as altro
    B, S, W = 8192, 20, 12
    pb = altro.problems.gen_random_linear_batch(B, n=12, m=4, N=50, steps=5 + S * W + 200, seed=1)
    mp, heat = altro.mpc.BatchMPC(pb), altro.mpc.BatchMPC(pb)
    for m_ in (mp, heat):
        m_.initial_solve()
        for i in range(5): m_.step(i)
    out = []
    for w in range(W):
        heat.run_async(100, first=5); heat.run_async(100, first=105); heat.synchronize()
        altro.timing_reset(mp.solver)
        mp.run_async(S, first=5 + w * S); mp.synchronize()
        out.append(float(altro.timing_get(mp.solver).sum()))
    print("%-16s" % sys.argv[1], " ".join("%6.2f" % x for x in out), " | mean %.3f ms" % (sum(out) / len(out)), flush=True)
else:
    for rep in range(2):
        for a in sys.argv[1:]:
            tag, lib = a.split("=", 1)
            e = dict(os.environ); e["ALTRO_HIP_LIB"] = lib
            subprocess.run([sys.executable, __file__, tag], env=e)
