"""Per (step, instance) comparison table of the horizon-N rocket MPC: GPU fused loop vs oracle."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "oracle"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
import oracle_py as O
import test_gpu_parity as T
theta, B, Nm, S, seed = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
tp, mp, orcs, oracle_step, Xt, Ut = T._rocket_track_mpc(O, theta, B, Nm, seed)
for i in range(S):
    mp.step(i)
    st, X, U = altro.stats(mp.solver), altro.states(mp.solver), altro.controls(mp.solver)
    at = altro.alpha_trace(mp.solver)
    for b in range(B):
        o = orcs[b]; oracle_step(b, i); so = o.solve()
        k = min(so.iterations, 16)
        same = np.array_equal(at[b, :k], np.array(so.alpha[:k]))
        print("step %2d inst %d: it %3d/%3d outer %d/%d st %d/%d alpha-same %d J %.8g/%.8g cmax %.3e/%.3e dX %.1e dU %.1e dJtr %.1e" % (
            i, b, st.iterations[b], so.iterations, st.iterations_outer[b], so.iterations_outer, st.status[b], so.status, same,
            st.cost[b], so.cost, st.c_max[b], so.c_max, T.rel_err(X[b], o.states()), T.rel_err(U[b], o.controls()),
            np.max(np.abs(st.cost_trace[b, :k] - np.array(so.J[:k])) / np.maximum(1, np.abs(np.array(so.J[:k]))))))
        if not same:
            print("     alpha g", at[b, :k], " o", np.array(so.alpha[:k]))
