"""PCIe-inclusive rate of the fine-grained ABI (the reference's own call sequence with host buffers
every step: random_linear_problem.jl:125-174): set_initial_state + set_reference + shift_fill +
solve + get_controls.  Reported in DESIGN.md section 4; never bench.py's `value`."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
B, S = 8192, 30
pb = altro.problems.gen_random_linear_batch(B, steps=S + 2)
prob = altro.mpc.gen_tracking_problem(pb)
sv = altro.ALTROSolver(prob, altro.SolverOptions(**altro.mpc.REF_OPTS))
altro.solve(sv)
N = pb.N
x = pb.Xtrack[:, 0].copy()
t_all = []
for i in range(S):
    U = altro.controls(sv)                       # device -> host
    t0 = time.perf_counter()
    x = np.einsum("bij,bj->bi", pb.A, x) + np.einsum("bij,bj->bi", pb.Bm, U[:, 0])
    x += pb.noise[i] * np.abs(x).max(axis=1, keepdims=True) / 100.0
    t_host = time.perf_counter() - t0
    t0 = time.perf_counter()
    altro.set_initial_state(sv, x)
    altro.update_trajectory(sv, np.ascontiguousarray(pb.Xtrack[:, i + 1:i + 1 + N]), np.ascontiguousarray(pb.Utrack[:, i + 1:i + N]))
    altro.shift_fill(sv, True, True)
    altro.solve(sv)
    U = altro.controls(sv)
    t_all.append(time.perf_counter() - t0)
st = altro.stats(sv)
t = np.median(t_all[3:])
print("fine-grained ABI, host buffers every step: median %.2f ms per step -> %.3g solves/s (batch %d); kernel %.2f ms; status ok %.4f" % (
    1e3 * t, B / t, B, st.tsolve_ms, (st.status == 1).mean()))
