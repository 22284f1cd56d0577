"""How many backward passes see the active set and penalty of the row's previous pass?  (needs the
-DALTRO_PHASE_STAMPS -DALTRO_DIAG_REUSE build: wave_cycles columns 3..6 carry the counts)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
pb = altro.problems.gen_random_linear_batch(B, steps=S + 5)
mp = altro.mpc.BatchMPC(pb)
mp.initial_solve()
for i in range(5): mp.step(i)
altro.timing_reset(mp.solver)
mp.run_async(S, first=5); mp.synchronize()
wc = altro.wave_cycles(mp.solver).astype(float)
ns, ni, nok = altro.solve_counters(mp.solver)
nb, nr, ntr = altro.work_counters(mp.solver)
print("solves %d, iterations %d, backward passes %d (rows)" % (ns.sum(), ni.sum(), nb.sum()))
print("first iteration of an inner solve: %d passes, %d with the previous pass's active set and penalty (%.3f)" % (wc[:, 4].sum(), wc[:, 3].sum(), wc[:, 3].sum() / wc[:, 4].sum()))
print("later iterations: %d passes, %d same (%.3f)" % (wc[:, 6].sum(), wc[:, 5].sum(), wc[:, 5].sum() / max(1, wc[:, 6].sum())))
print("lone passes (waves): %d" % wc[:, 7].sum())
# the slowest waves
order = np.argsort(-wc[:, 0])[:8]
for w in order:
    print("wave %5d: total %.2fM first %d/%d later %d/%d lone %d rows %s" % (w, wc[w, 0] / 1e6, wc[w, 3], wc[w, 4], wc[w, 5], wc[w, 6], wc[w, 7], ni.reshape(-1, 4)[w].tolist()))
