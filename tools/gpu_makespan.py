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
names = ["total", "bw4", "closed", "open", "todorov", "dual", "ls", "#lone", "bwlone", "fosweep", "adjoint", "#bw4", "#fo", "#aj", "#rc", "#ls"]
print("steps %d: per-instance iterations mean %.1f max %d; per-wave max-of-4 mean %.1f max %d" % (S, ni.mean(), ni.max(), ni.reshape(-1, 4).max(1).mean(), ni.max()))
print("wave cycles: mean %.2fM  p50 %.2fM  p99 %.2fM  max %.2fM  -> mean/max = %.3f" % (wcs[:, 0].mean() / 1e6, np.median(wcs[:, 0]) / 1e6, np.percentile(wcs[:, 0], 99) / 1e6, wcs[:, 0].max() / 1e6, wcs[:, 0].mean() / wcs[:, 0].max()))
fmt = lambda row: " ".join(("%s %.2fM" % (n, row[i] / 1e6)) if not n.startswith("#") else ("%s %.1f" % (n, row[i])) for i, n in enumerate(names))
print("mean wave  :", fmt(wcs.mean(0)))
for nm, ci, ti in (("four-row pass", 11, 1), ("lone pass", 7, 8), ("first-order sweep", 12, 9), ("costate sweep", 13, 10), ("closed rollout", 14, 2), ("trial sweep", 15, 6)):
    print("  %-18s %.0fk cycles each" % (nm, wcs[:, ti].sum() / max(1.0, wcs[:, ci].sum()) / 1e3))
st = altro.stats(mp.solver)
print("kernel ms %.2f" % st.tsolve_ms)
nb, nr, ntr = altro.work_counters(mp.solver)
ngc = altro.confirm_counter(mp.solver)
nfo = altro.reuse_counter(mp.solver)
print("per solve: iterations %.3f, backward passes %.3f, gains from memory %.3f, costate-confirmed %.3f, rollouts %.3f, trials %.3f; lone passes per wave %.1f" % (
    ni.sum() / ns.sum(), nb.sum() / ns.sum(), nfo.sum() / ns.sum(), ngc.sum() / ns.sum(), nr.sum() / ns.sum(), ntr.sum() / ns.sum(), wcs[:, 7].mean()))
# the slowest waves: which phase carries their extra time, and how many turns of the wave loop they took
print("(row iteration lists below name the instances 4w..4w+3: with grouping on, wave w holds OTHER instances -- run with ALTRO_NO_GROUP=1 for labels that match)")
order = np.argsort(-wcs[:, 0])[:5]
nit4 = ni.reshape(-1, 4)
for w in order:
    print("wave %5d: %s | row iterations %s" % (w, fmt(wcs[w]), nit4[w].tolist()))
tot_it = nit4.max(1)
print("cycles per wave-iteration (total / max-of-4 iterations): mean %.0fk, slowest five %s" % (
    (wcs[:, 0] / tot_it).mean() / 1e3, ", ".join("%.0fk" % (wcs[w, 0] / tot_it[w] / 1e3) for w in order)))
