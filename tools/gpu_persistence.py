"""How persistent is per-instance hardness between consecutive fused launches?  (Plain build; iteration counters only.)
Usage: gpu_persistence.py [steps=20] [batch=8192]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
pb = altro.problems.gen_random_linear_batch(B, steps=3 * S + 5)
mp = altro.mpc.BatchMPC(pb)
mp.initial_solve()
its = []
first = 0
for w, cnt in enumerate((5, S, S, S)):
    altro.timing_reset(mp.solver)
    mp.run_async(cnt, first=first); mp.synchronize()
    first += cnt
    ns, ni, nok = altro.solve_counters(mp.solver)
    its.append(ni.astype(float).copy())
    print("window %d (%d steps): iterations per instance mean %.1f p99 %.0f max %d; kernel ms %.2f" % (w, cnt, ni.mean(), np.percentile(ni, 99), ni.max(), altro.stats(mp.solver).tsolve_ms))
def rank(a): return np.argsort(np.argsort(a))
for a, b in ((0, 1), (1, 2), (2, 3)):
    x, y = its[a], its[b]
    rho = np.corrcoef(rank(x), rank(y))[0, 1]
    k = B // 20
    top_prev, top_next = set(np.argsort(-x)[:k]), set(np.argsort(-y)[:k])
    worst = np.argsort(-y)[:32]
    print("windows %d -> %d: Spearman %.3f; top-5%% overlap %.2f; of the 32 hardest instances of the next window, %d were in the previous window's top 5%%, %d in its top 20%%" % (
        a, b, rho, len(top_prev & top_next) / k, sum(1 for i in worst if i in top_prev), sum(1 for i in worst if i in set(np.argsort(-x)[:B // 5]))))
    w4 = y.reshape(-1, 4).max(1)
    print("   next window: max-of-4 per wave mean %.1f max %.0f; if rows were sorted by the previous window's count: mean %.1f max %.0f" % (
        w4.mean(), w4.max(), y[np.argsort(-x)].reshape(-1, 4).max(1).mean(), y[np.argsort(-x)].reshape(-1, 4).max(1).max()))
