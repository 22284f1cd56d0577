"""For the headline workload on the ORACLE: at every backward pass inside a solve, the highest knot whose second-order
cost expansion (active set, penalties) differs from the previous pass's.  Gains and cost-to-go Hessians above that knot
are the previous pass's, so a pass would only have to restart there (tools only: the oracle's orc_debug_pass_trace)."""
import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
import oracle_py
from helpers import make_oracle, mpc_update
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 25
pb = altro.problems.gen_random_linear_batch(B, n=12, m=4, N=50, steps=S + 1, seed=1)
L = oracle_py.lib()
L.orc_debug_pass_trace.restype = C.c_int
L.orc_debug_pass_trace.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
N = pb.N
rows = []      # (instance, step, pass index in the solve, kc)
iters = np.zeros(B, dtype=int)
for b in range(B):
    o = make_oracle(oracle_py, pb, b)
    buf = np.zeros(4096, dtype=np.int32)
    o.solve()
    for i in range(S):
        mpc_update(o, pb, b, i)
        L.orc_debug_pass_trace(o.h, buf.ctypes.data_as(C.POINTER(C.c_int)), 4096)
        o.solve()
        n = L.orc_debug_pass_trace(o.h, buf.ctypes.data_as(C.POINTER(C.c_int)), 4096)
        if i >= 5:
            iters[b] += o.stats().iterations
            for p in range(n):
                rows.append((b, i, p, int(buf[p])))
rows = np.array(rows)
first = rows[rows[:, 2] == 0]
later = rows[rows[:, 2] > 0]
print("instances %d, steps %d: passes %d (first of a solve %d, later %d)" % (B, S - 5, len(rows), len(first), len(later)))
def describe(name, r):
    if not len(r): return
    kc = r[:, 3]
    print("%-34s passes %5d | unchanged %.3f | knots a restarted pass walks / all: %.3f | kc quartiles %s" % (
        name, len(r), (kc < 0).mean(), ((kc + 1).clip(0, N - 1)).sum() / (len(r) * (N - 1)), np.percentile(kc, [25, 50, 75]).tolist()))
describe("first pass of a solve", first)
describe("later passes", later)
hard = np.argsort(-iters)[:max(1, B // 50)]
describe("later passes, hardest 2 %", later[np.isin(later[:, 0], hard)])
describe("first passes, hardest 2 %", first[np.isin(first[:, 0], hard)])
print("hardest instances: iterations in %d steps %s" % (S - 5, iters[hard].tolist()))
