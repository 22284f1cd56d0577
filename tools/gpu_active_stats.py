"""How often is the box active set of the headline workload empty / unchanged from one MPC solve to the next?
(feasibility check for reusing the gains of the previous solve's backward pass)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
B, S = 8192, 12
pb = altro.problems.gen_random_linear_batch(B, steps=S)
mp = altro.mpc.BatchMPC(pb)
mp.initial_solve()
prev = None
for i in range(S):
    mp.step(i)
    U = altro.controls(mp.solver)                       # (B, N-1, m)
    lam = altro.duals(mp.solver) if hasattr(altro, "duals") else None
    act = (np.abs(U) >= pb.u_bnd - 1e-9)
    if lam is not None:
        lam = np.asarray(lam)
        pos = (lam > 0).reshape(B, -1).any(1)
    else:
        pos = np.zeros(B, bool)
    empty = ~act.reshape(B, -1).any(1) & ~pos
    if prev is not None:
        same_shift = (act[:, :-1] == prev[:, 1:]).reshape(B, -1).all(1)    # active set moved with the trajectory
        same_knot = (act == prev).reshape(B, -1).all(1)                   # active set identical knot by knot
        print("step %d: empty active set %.3f; identical knot by knot to the previous solve %.3f; shifted copy %.3f; any dual > 0: %.3f" % (
            i, empty.mean(), same_knot.mean(), same_shift.mean(), pos.mean()))
    prev = act
