"""Where does the quadruped second-order-cone solve at N = 40 leave the oracle's iterate path?"""
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
B, S, N = 2048, 3, int(sys.argv[1]) if len(sys.argv) > 1 else 40
qp = P.gen_quadruped_problem(N=N, linearized_friction=False)
rng = np.random.default_rng(17)
t0 = rng.uniform(0.0, 0.8, B)
x0 = qp.x_des + rng.standard_normal((B, 12)) * np.array([.02, .02, .02, .05, .05, .05, .3, .3, .1, .3, .3, .3])
sample = list(range(0, B, 511)) + [B - 1]
A, Bm, d = np.zeros((B, S + N, 12, 12)), np.zeros((B, S + N, 12, 12)), np.zeros((B, S + N, 12))
cache = {}
for b in range(B):
    for t in range(S + N):
        c = tuple(P.trot_contacts(t0[b] + t * qp.dt))
        if c not in cache:
            cache[c] = P.quadruped_linearize(qp.x_des, np.zeros(12), qp.feet, np.array(c), qp.inertia, qp.mass, qp.dt)
        A[b, t], Bm[b, t], d[b, t] = cache[c]
noise = rng.standard_normal((S, B, 12))
sub = np.array(sample)
mp = _quadruped_device_loop(qp, x0[sub], A[sub], Bm[sub], d[sub], noise[:, sub], S, **({"strict": 1} if "strict" in sys.argv else {}))
mp.initial_solve()
orcs = [quadruped_oracle(O, qp, x0[b], A[b, :N - 1], Bm[b, :N - 1], d[b, :N - 1], P.QUADRUPED_OPTS) for b in sample]
sos = [o.solve() for o in orcs]
def cmp(tag):
    st, at = altro.stats(mp.solver), altro.alpha_trace(mp.solver)
    for q, (o, so) in enumerate(zip(orcs, sos)):
        k = min(so.iterations, 16)
        dj = np.abs(st.cost_trace[q, :k] - np.array(so.J[:k])) / np.maximum(1, np.abs(np.array(so.J[:k])))
        flag = "" if dj.max() < 1e-6 and st.iterations[q] == so.iterations else "   <-- differs"
        print("%s inst %d: it %d/%d status %d/%d max rel dJ %.1e%s" % (tag, sample[q], st.iterations[q], so.iterations, st.status[q], so.status, dj.max() if k else 0, flag))
        if flag:
            print("    J gpu  :", " ".join("%.8g" % v for v in st.cost_trace[q, :k]))
            print("    J orc  :", " ".join("%.8g" % v for v in so.J[:k]))
            print("    a gpu  :", " ".join("%.3g" % v for v in at[q, :k]))
            print("    a orc  :", " ".join("%.3g" % v for v in so.alpha[:k]))
            print("    cm gpu :", " ".join("%.3g" % v for v in st.cmax_trace[q, :k]))
            print("    cm orc :", " ".join("%.3g" % v for v in so.cmax_it[:k]))
cmp("cold")
for i in range(S):
    mp.step(i)
    for q, b in enumerate(sample):
        o = orcs[q]
        xn = o.plant_step() + 1e-3 * noise[i, b]
        o.set_dynamics(A[b, i + 1:i + N], Bm[b, i + 1:i + N], d[b, i + 1:i + N]); o.set_initial_state(xn); o.shift_fill(True, True)
        sos[q] = o.solve()
    cmp("tick %d" % i)
