"""A/B of builds of the library (tools/build_ab.sh, -DALTRO_DEV_WIDE_KERNEL=...) on one-wave-per-instance workloads: kernel time
of three 10-step windows and a checksum of the final states / controls / iteration counts (builds that only move code must agree
bit for bit).  Usage: gpu_wide_ab.py quad|n16|n32|n64|n12m6 tag=lib.so [tag=lib.so ...]"""
import sys, os, subprocess, hashlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
if len(sys.argv) == 3 and "=" not in sys.argv[2]:
    import numpy as np
    import altro_amd_loader
    import altro_mpc_icra2021_amd as altro
    from altro_mpc_icra2021_amd import problems as P, mpc as mpcm, api
    what, tag = sys.argv[1], sys.argv[2]
    W, K = 3, 10
    if what == "quad":
        B, N = 2048, 40
        qb = P.gen_quadruped_batch(B, N=N, steps=W + 3 * K, seed=17)
        qp, x0, A, Bm, d = qb.qp, qb.x0, qb.A, qb.Bm, qb.d
        Nt = W + 3 * K + N + 1
        prob = mpcm.quadruped_problem(qp, x0, A[:, :N - 1], Bm[:, :N - 1], d[:, :N - 1])
        mp = mpcm.TrackMPC(prob, api.SolverOptions(**P.QUADRUPED_OPTS), np.tile(qp.x_des, (B, Nt, 1)), np.zeros((B, Nt - 1, 12)),
                           qb.noise, (np.full(12, 1e-3),))
        api.set_dynamics_track(mp.solver, A, Bm, d, step_stride=1)
        api.initial_controls(mp.solver, np.tile(qp.u_hover, (B, N - 1, 1)))
    else:
        import re
        mm = re.match(r"n(\d+)(?:m(\d+))?", what)
        n, m = int(mm.group(1)), int(mm.group(2) or 4)
        B = int(os.environ.get("AB_BATCH", "8192" if n <= 32 else "2048"))
        pb = altro.problems.gen_random_linear_batch(B, n=n, m=m, N=50, steps=W + 3 * K, seed=5)
        mp = altro.mpc.BatchMPC(pb)
    mp.initial_solve()
    for i in range(W): mp.step(i)
    out = []
    for w in range(3):
        altro.timing_reset(mp.solver)
        mp.run_async(K, first=W + w * K); mp.synchronize()
        out.append(float(altro.timing_get(mp.solver).sum()))
    st = altro.stats(mp.solver)
    hsh = hashlib.sha1(altro.states(mp.solver).tobytes() + altro.controls(mp.solver).tobytes() + st.iterations.tobytes() + st.status.tobytes()).hexdigest()[:12]
    print("%-10s %-12s" % (what, tag), " ".join("%8.2f" % x for x in out), "ms | sum %.1f | iters %d | %s" % (sum(out), int(st.iterations.sum()), hsh), flush=True)
else:
    what = sys.argv[1]
    for rep in range(2):
        for a in sys.argv[2:]:
            tag, lib = a.split("=", 1)
            e = dict(os.environ); e["ALTRO_HIP_LIB"] = lib
            subprocess.run([sys.executable, __file__, what, tag], env=e)
