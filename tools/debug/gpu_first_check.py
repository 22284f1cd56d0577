"""Ad-hoc first GPU check: batched HIP solve vs CPU oracle on the same inputs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
import oracle_py as O
from helpers import make_oracle, mpc_update, REF_OPTS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = 6
pb = altro.problems.gen_random_linear_batch(B, steps=S)
mp = altro.mpc.BatchMPC(pb)
t0 = time.time(); mp.initial_solve(); print("initial solve wall", time.time() - t0)
st = altro.stats(mp.solver)
print("gpu cold: iters", st.iterations[:8], "status", st.status[:8], "cost", st.cost[:4], "ms", st.tsolve_ms)
orcs = [make_oracle(O, pb, b) for b in range(min(B, 8))]
for o in orcs: o.solve()
for i in range(S):
    mp.step(i)
    st = altro.stats(mp.solver)
    X = altro.states(mp.solver); U = altro.controls(mp.solver)
    x0g = mp.x0()
    errs = []
    for b, o in enumerate(orcs):
        x0 = mpc_update(o, pb, b, i)
        so = o.solve()
        ex = np.abs(o.states() - X[b]).max(); eu = np.abs(o.controls() - U[b]).max()
        errs.append((float(np.abs(x0 - x0g[b]).max()), ex, eu, so.iterations, int(st.iterations[b]), so.status, int(st.status[b]),
                     so.cost, float(st.cost[b])))
    print("step", i, "ms", round(st.tsolve_ms, 3))
    for e in errs: print("   x0err %.1e Xerr %.1e Uerr %.1e it(o,g) %d %d status %d %d cost %.10g %.10g" % e)
