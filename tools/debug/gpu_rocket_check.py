"""Ad-hoc GPU check of the conic path: rocket landing cold solve vs the CPU oracle."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "oracle"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
from altro_mpc_icra2021_amd import problems as P
import oracle_py as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 61
B = 6
rp = P.gen_rocket_problem(N=N, tf=15.0, Qfk=1e4, Rk=1.0, theta_thrust_max=5.0, theta_glideslope=45.0)
rng = np.random.default_rng(0)
x0 = np.tile(rp.x0, (B, 1)) + rng.standard_normal((B, 6)) * np.array([1, 1, 1, .3, .3, .3]) * np.linspace(0, 1, B)[:, None]
opts = dict(cost_tolerance_intermediate=1e-4, penalty_scaling=500., penalty_initial=1e-2, constraint_tolerance=1e-5,
            iterations=5000, iterations_inner=100, iterations_linesearch=100, iterations_outer=60)
n, m = rp.n, rp.m
model = altro.LinearModel(rp.A, rp.Bm, rp.f, dt=rp.dt)
Xr = np.tile(rp.xf, (B, N, 1)); Ur = np.zeros((B, N - 1, m))
obj = altro.TrackingObjective(rp.Q, rp.R, rp.Qf, Xr, Ur)
cons = altro.ConstraintList(n, m, N)
for c in rp.constraints:
    con = altro.NormConstraint(c.A, c.b) if c.kind == P.SOC else altro.LinearConstraint(c.A, c.b, equality=(c.sense == P.EQ))
    cons.add_constraint(con, (c.k_first + 1, c.k_last + 1))
prob = altro.Problem(model, obj, cons, x0=x0, N=N, U0=np.tile(rp.U0, (B, 1, 1)))
sv = altro.ALTROSolver(prob, altro.SolverOptions(**opts))
t0 = time.time(); altro.solve(sv); print("gpu solve wall", time.time() - t0)
st = altro.stats(sv)
X, U = altro.states(sv), altro.controls(sv)
for b in range(B):
    s = O.OracleSolver(n, m, N, rp.dt)
    s.set_dynamics(rp.A, rp.Bm, rp.f); s.set_cost(rp.Q, rp.R, rp.Qf)
    s.set_reference(Xr[b], Ur[b]); s.set_initial_state(x0[b]); s.set_controls(rp.U0)
    for c in rp.constraints:
        s.add_affine(c.kind, c.sense, c.A, c.b, c.k_first, c.k_last)
    s.set_opts(O.default_opts(**opts))
    so = s.solve()
    print("inst %d  iters o/g %d %d  outer %d %d  status %d %d  J %.10g %.10g  cmax %.3e %.3e  Xerr %.2e Uerr %.2e" % (
        b, so.iterations, st.iterations[b], so.iterations_outer, st.iterations_outer[b], so.status, st.status[b],
        so.cost, st.cost[b], so.c_max, st.c_max[b], np.abs(s.states() - X[b]).max(), np.abs(s.controls() - U[b]).max()))
    k = min(so.iterations, 16)
    jo = np.array(so.J[:k]); jg = st.cost_trace[b, :k]
    print("      J trace rel err", np.abs(jo - jg).max() / np.abs(jo).max(), " first J o/g", jo[:3], jg[:3])
