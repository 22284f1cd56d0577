"""How predictable is an instance's iteration count?  Correlation of per-instance AL-iLQR iterations
between consecutive windows of MPC steps (random_linear_mpc, batch 8192)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
B = 8192
pb = altro.problems.gen_random_linear_batch(B, steps=110)
mp = altro.mpc.BatchMPC(pb)
mp.initial_solve()
def iters():
    return altro.solve_counters(mp.solver)[1].astype(np.int64).copy()
c0 = iters()
wins = [(0, 5), (5, 15), (15, 55), (55, 105)]
got = []
for a, b in wins:
    mp.run_async(b - a, first=a); mp.synchronize()
    c1 = iters(); got.append(c1 - c0); c0 = c1
for i in range(len(wins)):
    print("steps %s: iterations mean %.2f/step  sd over instances %.2f  max %.2f" % (wins[i], got[i].mean() / (wins[i][1] - wins[i][0]), got[i].std() / (wins[i][1] - wins[i][0]), got[i].max() / (wins[i][1] - wins[i][0])))
for i in range(len(wins)):
    for k in range(i + 1, len(wins)):
        print("corr(steps %s, steps %s) = %.3f" % (wins[i], wins[k], np.corrcoef(got[i], got[k])[0, 1]))
tot = got[1] + got[2] + got[3]
print("corr(warmup 0-5, steps 5-105) = %.3f" % np.corrcoef(got[0], tot)[0, 1])
# how good would a sort by the warm-up count be?  per-wave max-of-4 with and without sorting
def wave_max(order):
    return tot[order].reshape(-1, 4).max(1)
ident = np.arange(B)
by_warm = np.argsort(got[0], kind="stable")
by_oracle = np.argsort(tot, kind="stable")
for name, o in (("as given", ident), ("sorted by warm-up", by_warm), ("sorted by the truth", by_oracle)):
    w = wave_max(o)
    print("%-20s per-wave max-of-4: mean %.1f  max %d" % (name, w.mean(), w.max()))
np.save(os.path.join(R, "gpurun_out", "predict_iters.npy"), np.stack(got))
