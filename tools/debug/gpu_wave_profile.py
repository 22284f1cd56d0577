"""Wave time and phase mix by position in the grouped order (diagnostic build): python tools/debug/gpu_wave_profile.py [steps] [batch]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
w = altro.wave_cycles(mp.solver).astype(float)
nw = w.shape[0]
print("kernel ms %.2f; waves %d" % (altro.stats(mp.solver).tsolve_ms, nw))
print("wave-index decile: total  bw4  bwlone  closed  open  fosweep  adjoint  (M ticks)  #bw4 #lone #fo #aj #rc")
for d in range(16):
    s = slice(d * nw // 16, (d + 1) * nw // 16)
    m = w[s].mean(0)
    print("%2d  %6.2f  %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f   %5.1f %5.1f %5.1f %5.1f %5.1f   max %.2f" % (d, m[0] / 1e6, m[1] / 1e6, m[8] / 1e6, m[2] / 1e6, m[3] / 1e6, m[9] / 1e6, m[10] / 1e6, m[11], m[7], m[12], m[13], m[14], w[s, 0].max() / 1e6))
h, e = np.histogram(w[:, 0] / 1e6, bins=20)
print("histogram of wave totals (M ticks):", " ".join("%.1f:%d" % (e[i], h[i]) for i in range(len(h))))
