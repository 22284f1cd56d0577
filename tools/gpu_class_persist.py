"""Is 'this instance needs backward passes' persistent from one 20-step launch to the next?  (grouping instances by it
puts the rows that take their gains from memory into the same waves)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
B, S, L = 8192, 20, 5
pb = altro.problems.gen_random_linear_batch(B, steps=S * L + 5)
mp = altro.mpc.BatchMPC(pb)
mp.initial_solve()
for i in range(5): mp.step(i)
prev = altro.work_counters(mp.solver)[0].copy()
got = []
for l in range(L):
    mp.run_async(S, first=5 + l * S); mp.synchronize()
    nb = altro.work_counters(mp.solver)[0].copy()
    got.append(nb - prev); prev = nb
got = np.array(got)
for l in range(L):
    g = got[l]
    print("launch %d: passes per instance mean %.2f; instances with 0 passes %.3f, <= 2: %.3f, >= 15: %.3f" % (l, g.mean(), (g == 0).mean(), (g <= 2).mean(), (g >= 15).mean()))
for l in range(1, L):
    a, b = got[l - 1], got[l]
    print("corr(launch %d, %d) = %.3f;  P(next <= 2 | this <= 2) = %.3f;  P(next >= 10 | this >= 10) = %.3f" % (l - 1, l, np.corrcoef(a, b)[0, 1], ((a <= 2) & (b <= 2)).sum() / max(1, (a <= 2).sum()), ((a >= 10) & (b >= 10)).sum() / max(1, (a >= 10).sum())))
# what a sort by the previous launch's count would give: per-wave max-of-4 passes (a proxy of the passes the wave runs)
for l in range(1, L):
    order = np.argsort(got[l - 1], kind="stable")
    w0 = got[l].reshape(-1, 4).max(1)
    w1 = got[l][order].reshape(-1, 4).max(1)
    wt = np.sort(got[l]).reshape(-1, 4).max(1)
    print("launch %d: per-wave max-of-4 passes: as given mean %.1f, sorted by the previous launch %.1f, sorted by the truth %.1f" % (l, w0.mean(), w1.mean(), wt.mean()))
