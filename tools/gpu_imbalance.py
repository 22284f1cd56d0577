"""How persistent is per-instance difficulty?  Per-instance iteration totals over a fused run."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
B, S = 8192, 43
pb = altro.problems.gen_random_linear_batch(B, steps=S)
mp = altro.mpc.BatchMPC(pb)
mp.initial_solve()
for i in range(3): mp.step(i)
altro.timing_reset(mp.solver)
mp.run_async(20, first=3); mp.synchronize()
ns, ni1, nok = altro.solve_counters(mp.solver)
nb1, nr1, nt1 = altro.work_counters(mp.solver)
altro.timing_reset(mp.solver)
mp.run_async(20, first=23); mp.synchronize()
ns, ni2, nok = altro.solve_counters(mp.solver)
nb2, nr2, nt2 = altro.work_counters(mp.solver)
print("per-instance iterations over 20 steps: mean %.1f median %.0f p99 %.0f max %d" % (ni1.mean(), np.median(ni1), np.percentile(ni1, 99), ni1.max()))
print("correlation of per-instance totals between consecutive 20-step windows: %.3f" % np.corrcoef(ni1, ni2)[0, 1])
w1 = ni1.reshape(-1, 4); 
print("per-wave sum-of-max proxy: mean of max-over-4 totals %.1f ; mean instance %.1f" % (w1.max(1).mean(), ni1.mean()))
# if instances were sorted by window-1 difficulty and grouped, what would window-2 look like?
order = np.argsort(ni1)
w2s = ni2[order].reshape(-1, 4)
w2 = ni2.reshape(-1, 4)
print("window 2: unsorted waves: mean(max4)=%.1f max=%d ; sorted by window-1 totals: mean(max4)=%.1f max=%d" % (w2.max(1).mean(), w2.max(), w2s.max(1).mean(), w2s.max()))
hard = ni1 >= np.percentile(ni1, 95)
print("top-5%% instances of window 1 account for %.1f%% of window-2 iterations; their mean %.1f vs others %.1f" % (100 * ni2[hard].sum() / ni2.sum(), ni2[hard].mean(), ni2[~hard].mean()))
print("rollouts/solve %.2f trials/solve %.2f backward/solve %.2f" % (nr2.sum() / (B * 20), nt2.sum() / (B * 20), nb2.sum() / (B * 20)))
wc = altro.wave_cycles(mp.solver)[:, 0].astype(float)
tot4 = w2.max(1)
print("corr(wave cycles, max-of-4 iteration totals) = %.3f ; corr(wave cycles, sum-of-4) = %.3f" % (np.corrcoef(wc, tot4)[0, 1], np.corrcoef(wc, w2.sum(1))[0, 1]))
print("wave cycles: mean %.3g max %.3g ; slowest wave instance totals:" % (wc.mean(), wc.max()), w2[np.argmax(wc)])
wcs = altro.wave_cycles(mp.solver).astype(float)
names = ["total", "backward", "closed", "open", "todorov", "dual", "ls"]
print("mean wave  :", " ".join("%s %.2fM" % (n, wcs[:, i].mean() / 1e6) for i, n in enumerate(names)))
idx = np.argsort(-wcs[:, 0])[:6]
for w in idx:
    print("slow wave %4d:" % w, " ".join("%s %.2fM" % (n, wcs[w, i] / 1e6) for i, n in enumerate(names)), "iters", w2[w], "bw", nb2.reshape(-1,4)[w], "ro", nr2.reshape(-1,4)[w], "tr", nt2.reshape(-1,4)[w])
idx = np.argsort(wcs[:, 0])[:3]
for w in idx:
    print("fast wave %4d:" % w, " ".join("%s %.2fM" % (n, wcs[w, i] / 1e6) for i, n in enumerate(names)), "iters", w2[w], "bw", nb2.reshape(-1,4)[w], "ro", nr2.reshape(-1,4)[w], "tr", nt2.reshape(-1,4)[w])
# per-XCD (block % 8) mean
print("mean total by blockIdx%8:", [round(wcs[b::8, 0].mean() / 1e6, 2) for b in range(8)])
