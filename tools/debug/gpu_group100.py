"""100-step fused launches with and without grouping (ALTRO_GROUP_MAX_STEPS), four windows each, clocks up."""
import sys, os, subprocess
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
if len(sys.argv) > 1:
    import altro_amd_loader
    import altro_mpc_icra2021_amd as altro
    B, S, W = 8192, 100, 4
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
    print("%-26s" % sys.argv[1], " ".join("%6.2f" % x for x in out), " | mean %.2f ms" % (sum(out) / len(out)), flush=True)
else:
    for rep in range(2):
        for tag, env in (("not grouped (default)", {}), ("grouped", {"ALTRO_GROUP_MAX_STEPS": "128"})):
            e = dict(os.environ); e.update(env)
            subprocess.run([sys.executable, __file__, tag], env=e)
