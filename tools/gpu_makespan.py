"""Makespan vs mean wave time of the fused 100-step launch (needs the -DALTRO_PHASE_STAMPS build)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
pb = altro.problems.gen_random_linear_batch(B, steps=S + 5)
mp = altro.mpc.BatchMPC(pb)
mp.initial_solve()
for i in range(5): mp.step(i)
altro.timing_reset(mp.solver)
mp.run_async(S, first=5); mp.synchronize()
ns, ni, nok = altro.solve_counters(mp.solver)
wcs = altro.wave_cycles(mp.solver).astype(float)
names = ["total", "backward", "closed", "open", "todorov", "dual", "ls"]
print("steps %d: per-instance iterations mean %.1f max %d; per-wave max-of-4 mean %.1f max %d" % (S, ni.mean(), ni.max(), ni.reshape(-1, 4).max(1).mean(), ni.max()))
print("wave cycles: mean %.2fM  p50 %.2fM  p99 %.2fM  max %.2fM  -> mean/max = %.3f" % (wcs[:, 0].mean() / 1e6, np.median(wcs[:, 0]) / 1e6, np.percentile(wcs[:, 0], 99) / 1e6, wcs[:, 0].max() / 1e6, wcs[:, 0].mean() / wcs[:, 0].max()))
print("mean wave  :", " ".join("%s %.2fM" % (n, wcs[:, i].mean() / 1e6) for i, n in enumerate(names)))
st = altro.stats(mp.solver)
print("kernel ms %.2f" % st.tsolve_ms)
nb, nr, ntr = altro.work_counters(mp.solver)
ngc = altro.confirm_counter(mp.solver)
nbw = nb.reshape(-1, 4).max(1).astype(float)      # backward passes a wave ran = those of its busiest row (roughly)
print("per wave: backward passes (max row) mean %.1f -> %.0fk cycles per pass, %.0f per knot; costate sweeps mean %.1f -> %.0fk cycles each" % (
    nbw.mean(), wcs[:, 1].mean() / nbw.mean() / 1e3, wcs[:, 1].mean() / nbw.mean() / (pb.N - 1),
    ngc.reshape(-1, 4).max(1).mean(), wcs[:, 4].mean() / max(1.0, ngc.reshape(-1, 4).max(1).mean()) / 1e3))
# the slowest waves: which phase carries their extra time, and how many turns of the wave loop they took
order = np.argsort(-wcs[:, 0])[:5]
nit4 = ni.reshape(-1, 4)
for w in order:
    print("wave %5d: %s | row iterations %s" % (w, " ".join("%s %.2fM" % (n, wcs[w, i] / 1e6) for i, n in enumerate(names)), nit4[w].tolist()))
tot_it = nit4.max(1)
print("cycles per wave-iteration (total / max-of-4 iterations): mean %.0fk, slowest five %s" % (
    (wcs[:, 0] / tot_it).mean() / 1e3, ", ".join("%.0fk" % (wcs[w, 0] / tot_it[w] / 1e3) for w in order)))
