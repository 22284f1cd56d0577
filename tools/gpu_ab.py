"""A/B of the scheduling switches on one random-linear shape: python tools/gpu_ab.py n m N B steps"""
import sys, os, subprocess, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
if len(sys.argv) > 6:
    import numpy as np, time
    import altro_amd_loader
    import altro_mpc_icra2021_amd as altro
    n, m, N, B, S = map(int, sys.argv[1:6])
    HEAT = 300   # MPC steps run right before the timed launch: the clocks of an idle GPU take tens of ms to come up
    pb = altro.problems.gen_random_linear_batch(B, n=n, m=m, N=N, steps=S + 5 + HEAT, seed=10)
    mp = altro.mpc.BatchMPC(pb)
    mp.initial_solve()
    for i in range(5): mp.step(i)
    for h in range(0, HEAT, 100): mp.run_async(100, first=5 + h)
    mp.synchronize()
    altro.timing_reset(mp.solver)
    mp.run_async(S, first=5 + HEAT); mp.synchronize()
    ms = altro.timing_get(mp.solver)
    nb = altro.work_counters(mp.solver)[0]; nfo = altro.reuse_counter(mp.solver); ngc = altro.confirm_counter(mp.solver)
    print("%-34s %.3f ms  %.2f M solves/s  passes/solve %.3f reuse %.3f confirm %.3f" % (sys.argv[6], ms.sum(), B * S / ms.sum() / 1e3, nb.sum() / (B * S), nfo.sum() / (B * S), ngc.sum() / (B * S)))
else:
    for tag, env in (("default", {}), ("no reuse", {"ALTRO_NO_REUSE": "1"}), ("no reuse, no lone", {"ALTRO_NO_REUSE": "1", "ALTRO_NO_LONE": "1"}),
                     ("no group", {"ALTRO_NO_GROUP": "1"}), ("group mode 2 (alternating)", {"ALTRO_GROUP_MODE": "2"}), ("group mode 3 (balanced)", {"ALTRO_GROUP_MODE": "3"}), ("no resync", {"ALTRO_NO_RESYNC": "1"}), ("no lone", {"ALTRO_NO_LONE": "1"}),
                     ("nothing (round 2 schedule)", {"ALTRO_NO_REUSE": "1", "ALTRO_NO_LONE": "1", "ALTRO_NO_RESYNC": "1", "ALTRO_NO_GROUP": "1"})):
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, __file__] + sys.argv[1:6] + [tag], env=e)
