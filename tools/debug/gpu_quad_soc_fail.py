"""Do the quadruped second-order-cone solves that end at the cost limit do so in the oracle too?"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "oracle"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
from altro_mpc_icra2021_amd import problems as P
import oracle_py as O
from helpers import quadruped_oracle
from test_gpu_parity import _quadruped_device_loop
B, S, N = 2048, 3, 40
qp = P.gen_quadruped_problem(N=N, linearized_friction=False)
rng = np.random.default_rng(17)
t0 = rng.uniform(0.0, 0.8, B)
x0 = qp.x_des + rng.standard_normal((B, 12)) * np.array([.02, .02, .02, .05, .05, .05, .3, .3, .1, .3, .3, .3])
A, Bm, d = np.zeros((B, S + N, 12, 12)), np.zeros((B, S + N, 12, 12)), np.zeros((B, S + N, 12))
cache = {}
for b in range(B):
    for t in range(S + N):
        c = tuple(P.trot_contacts(t0[b] + t * qp.dt))
        if c not in cache:
            cache[c] = P.quadruped_linearize(qp.x_des, np.zeros(12), qp.feet, np.array(c), qp.inertia, qp.mass, qp.dt)
        A[b, t], Bm[b, t], d[b, t] = cache[c]
noise = rng.standard_normal((S, B, 12))
mp = _quadruped_device_loop(qp, x0, A, Bm, d, noise, S)
mp.initial_solve()
sts = [altro.stats(mp.solver)]
for i in range(S):
    mp.step(i); sts.append(altro.stats(mp.solver))
bad = sorted(set(np.nonzero(np.any([s.status != 1 for s in sts], axis=0))[0].tolist()))
print("instances with a failed solve:", len(bad), bad[:20])
for b in bad[:6]:
    o = quadruped_oracle(O, qp, x0[b], A[b, :N - 1], Bm[b, :N - 1], d[b, :N - 1], P.QUADRUPED_OPTS)
    so = o.solve(); line = ["%d/%d it %d/%d" % (sts[0].status[b], so.status, sts[0].iterations[b], so.iterations)]
    for i in range(S):
        xn = o.plant_step() + 1e-3 * noise[i, b]
        o.set_dynamics(A[b, i + 1:i + N], Bm[b, i + 1:i + N], d[b, i + 1:i + N]); o.set_initial_state(xn); o.shift_fill(True, True)
        so = o.solve(); line.append("%d/%d it %d/%d" % (sts[i + 1].status[b], so.status, sts[i + 1].iterations[b], so.iterations))
    print("instance %d: GPU/oracle status and iterations per solve: %s" % (b, "; ".join(line)))
