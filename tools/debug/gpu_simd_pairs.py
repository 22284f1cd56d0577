"""Which waves of the fused headline launch share a SIMD?  (diagnostic build: wave_cycles[4] = SIMD key)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
S, B = 20, 8192
pb = altro.problems.gen_random_linear_batch(B, steps=S + 5)
mp = altro.mpc.BatchMPC(pb)
mp.initial_solve()
for i in range(5): mp.step(i)
mp.run_async(S, first=5); mp.synchronize()
wc = altro.wave_cycles(mp.solver)
key = wc[:, 4]
tot = wc[:, 0].astype(float)
from collections import defaultdict
g = defaultdict(list)
for w, k in enumerate(key): g[int(k)].append(w)
sizes = np.bincount([len(v) for v in g.values()])
print("distinct SIMDs %d; waves per SIMD histogram %s" % (len(g), sizes.tolist()))
pairs = [v for v in g.values() if len(v) == 2]
d = np.array([abs(a - b) for a, b in pairs])
print("block-index distance of the two waves of a SIMD: min %d median %d max %d; most common %s" % (d.min(), np.median(d), d.max(), np.bincount(d).argsort()[-5:][::-1].tolist()))
s = np.array([tot[a] + tot[b] for a, b in pairs]) / 1e6
m = np.array([max(tot[a], tot[b]) for a, b in pairs]) / 1e6
print("per SIMD: sum of its two waves' cycles mean %.1fM max %.1fM; longer wave mean %.1fM max %.1fM" % (s.mean(), s.max(), m.mean(), m.max()))
xs = np.array([[a, b] for a, b in pairs[:12]])
print("examples", xs.tolist())
