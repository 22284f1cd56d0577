"""Per-turn trace of one wave of the fused headline launch (diagnostic build + ALTRO_DEBUG_TRACE_WAVE=w)."""
import sys, os, ctypes as C
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
s = mp.solver
n = B // 4 * 16 + 4096
buf = np.zeros(n, dtype=np.int64); cnt = C.c_int32(0)
s._chk(s._L.altro_batch_get_wave_cycles(s.h, buf.ctypes.data_as(C.POINTER(C.c_int64)), n, C.byref(cnt)))
tr = buf[B // 4 * 16:].reshape(1024, 4)
w = int(os.environ["ALTRO_DEBUG_TRACE_WAVE"])
wc = buf[:B // 4 * 16].reshape(-1, 16)
print("wave", w, "total %.2fM" % (wc[w, 0] / 1e6), "#bw4 %d #lone %d #rc %d" % (wc[w, 11], wc[w, 7], wc[w, 14]))
ph = "BOID"
for t in range(1, 200):
    row = tr[t]
    if not row.any(): break
    print("%3d %6.2fM " % (t, (int(row[0]) >> 32) * 1024 / 1e6) + "  ".join("%s it%d/%d st%2d o%d" % (ph[c & 15], (c >> 4) & 255, (c >> 12) & 255, (c >> 20) & 255, (c >> 28) & 15) for c in (int(x) & 0xFFFFFFFF for x in row)))
