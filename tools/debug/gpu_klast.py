"""Where is the LAST knot with an active bound at the end of every MPC step of the headline workload, and how many
backward passes did that step take?  A pass only has to recompute the second-order part from that knot down: above it the
active set is empty and the gains are those of the unconstrained problem (a per-instance table)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
B, S = 8192, 25
pb = altro.problems.gen_random_linear_batch(B, n=12, m=4, N=int(os.environ.get("KN", 50)), steps=S + 1)
mp = altro.mpc.BatchMPC(pb)
mp.initial_solve()
N = pb.N
tot_pass = np.zeros(B, dtype=np.int64)
knots_all, knots_needed = 0, 0
hist = np.zeros(N + 1, dtype=np.int64)
per_inst_needed = np.zeros(B)
prev = altro.work_counters(mp.solver)[0].copy()
for i in range(S):
    mp.step(i)
    nb = altro.work_counters(mp.solver)[0].copy()
    p = nb - prev; prev = nb
    U = altro.controls(mp.solver)                       # [B][N-1][m]
    lam = altro.get_duals(mp.solver, 0)                 # [B][nk][2][n+m]
    act_u = (np.abs(U) >= pb.u_bnd - 1e-12).any(-1)      # [B][N-1]
    act_l = (lam > 0).any(-1).any(-1)[:, :N - 1]
    act = act_u | act_l
    klast = np.where(act.any(1), (N - 2) - np.argmax(act[:, ::-1], axis=1), -1)    # -1: nothing active
    if i >= 5:
        tot_pass += p
        knots_all += (p * (N - 1)).sum()
        knots_needed += (p * (klast + 1)).sum()
        per_inst_needed += p * (klast + 1)
        np.add.at(hist, klast + 1, p)
print("passes per solve %.3f; pass knots needed / all = %.3f" % (tot_pass.mean() / (S - 5), knots_needed / knots_all))
print("passes by (last active knot + 1):", hist.tolist())
order = np.argsort(-tot_pass)
for frac in (0.001, 0.01, 0.05, 0.25):
    sel = order[:max(1, int(B * frac))]
    print("hardest %.1f %% instances: passes/solve %.2f; knots needed / all = %.3f" % (frac * 100, tot_pass[sel].mean() / (S - 5), per_inst_needed[sel].sum() / (tot_pass[sel].sum() * (N - 1))))
