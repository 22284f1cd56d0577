"""Gains of one cold solve, two builds of the library (separate processes): bitwise comparison.  gpu_gains_ab.py n m tagA=libA tagB=libB"""
import sys, os, subprocess, hashlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
if "=" not in sys.argv[-1]:
    import numpy as np
    import altro_amd_loader
    import altro_mpc_icra2021_amd as altro
    n, m = int(sys.argv[1]), int(sys.argv[2])
    pb = altro.problems.gen_random_linear_batch(8, n=n, m=m, N=21, steps=2, seed=3)
    mp = altro.mpc.BatchMPC(pb, altro.SolverOptions(strict=1, **altro.mpc.REF_OPTS)) if hasattr(altro.mpc, "REF_OPTS") else altro.mpc.BatchMPC(pb)
    mp.initial_solve()
    K, d = altro.gains(mp.solver)
    np.save(sys.argv[3], np.concatenate([K.ravel(), d.ravel(), altro.states(mp.solver).ravel()]))
else:
    n, m = sys.argv[1], sys.argv[2]
    outs = []
    for a in sys.argv[3:]:
        tag, lib = a.split("=", 1)
        e = dict(os.environ); e["ALTRO_HIP_LIB"] = lib
        f = "/tmp/gains_%s.npy" % tag
        subprocess.run([sys.executable, __file__, n, m, f], env=e, check=True)
        outs.append(f)
    import numpy as np
    a, b = np.load(outs[0]), np.load(outs[1])
    print("equal:", np.array_equal(a, b), "max abs diff %.3e" % np.abs(a - b).max(), "differing entries", int((a != b).sum()), "of", a.size)
