"""Throughput of the one-wave-per-instance kernel (solve_wide.h) on the shapes the 16-lane kernel
cannot hold: BASELINE configs[3] (state-dimension sweep, m = 4, N = 50, 8192 instances per GPU)
and configs[4] (quadruped N = 40, 2048 instances per GPU).  Not bench.py lines: DESIGN.md section 4."""
import sys, os, time, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "tests")):
    sys.path.insert(0, p)
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
from altro_mpc_icra2021_amd import problems as P


def flops_backward(n, m, N):
    return (N - 1) * (4 * n ** 3 + 8 * n * n * m + 6 * n * m * m + m ** 3 / 3 + 2 * n * n + 8 * n * m + 4 * m * m)


def flops_forward(n, m, N):
    return (N - 1) * (2 * n * n + 4 * n * m + 2 * (n + m)) + N * 3 * (n + m)


def sweep(n, m, N, B, S):
    pb = altro.problems.gen_random_linear_batch(B, n=n, m=m, N=N, steps=S + 3, seed=10)
    mp = altro.mpc.BatchMPC(pb)
    mp.initial_solve()
    for i in range(3):
        mp.step(i)
    altro.timing_reset(mp.solver)
    t0 = time.perf_counter()
    mp.run_async(S, first=3)
    mp.synchronize()
    dt = time.perf_counter() - t0
    ns, ni, nok = altro.solve_counters(mp.solver)
    nb, nr, ntr = altro.work_counters(mp.solver)
    fl = nb.sum() * flops_backward(n, m, N) + (nr.sum() + ntr.sum()) * flops_forward(n, m, N)
    print(json.dumps({"workload": "random_linear_mpc n=%d m=%d N=%d batch=%d (wide kernel)" % (n, m, N, B), "steps": S,
                      "solves_per_s": B * S / dt, "ms_per_step": 1e3 * dt / S, "iterations_mean": float(ni.sum() / ns.sum()),
                      "succeeded_frac": float(nok.sum() / ns.sum()), "tflops": fl / dt / 1e12}), flush=True)


def quadruped(N, B, S):
    from helpers import quadruped_gpu_problem
    qp = P.gen_quadruped_problem(N=N)
    rng = np.random.default_rng(7)
    phases = rng.uniform(0.0, 0.8, 16)
    D = [[qp.dynamics(ph + i * qp.dt) for ph in phases] for i in range(S + 1)]      # 16 gait phases, tiled over the batch

    def dyn(i):
        idx = np.arange(B) % 16
        return (np.stack([D[i][j][0] for j in range(16)])[idx], np.stack([D[i][j][1] for j in range(16)])[idx],
                np.stack([D[i][j][2] for j in range(16)])[idx])
    x0 = qp.x_des + rng.standard_normal((B, 12)) * np.array([.02, .02, .02, .05, .05, .05, .3, .3, .1, .3, .3, .3])
    A, Bm, d = dyn(0)
    sv = altro.ALTROSolver(quadruped_gpu_problem(altro, qp, x0, A, Bm, d), altro.SolverOptions(**P.QUADRUPED_OPTS))
    altro.solve(sv)
    tk, its, ok = [], [], []
    for i in range(1, S + 1):
        X, U = altro.states(sv), altro.controls(sv)
        xn = X[:, 1] + 1e-3 * rng.standard_normal((B, 12))
        A, Bm, d = dyn(i)
        altro.set_dynamics(sv, altro.LinearModel(A, Bm, d, dt=qp.dt, per_knot=True))
        altro.set_initial_state(sv, xn)
        altro.shift_fill(sv, True, True)
        altro.solve(sv)
        st = altro.stats(sv)
        tk.append(st.tsolve_ms); its.append(st.iterations.mean()); ok.append((st.status == 1).mean()); mx = max(locals().get("mx", 0), int(st.iterations.max()))
    t = np.median(tk)
    print(json.dumps({"workload": "quadruped trot MPC n=12 m=12 N=%d batch=%d (wide kernel, per-knot dynamics re-uploaded every step)" % (N, B),
                      "steps": S, "kernel_ms_per_step": float(t), "kernel_ms_per_step_min": float(np.min(tk)), "solves_per_s_kernel": B / t * 1e3,
                      "solves_per_s_kernel_best_step": B / float(np.min(tk)) * 1e3,
                      "iterations_mean": float(np.mean(its)), "iterations_max": mx, "succeeded_frac": float(np.mean(ok))}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "sweep"):
        for n in (16, 32, 48, 64):      # n = 8 (m = 4) runs on the 16-lane kernel
            sweep(n, 4, 50, 8192 if n <= 32 else 2048, 10)
    if which in ("all", "horizon"):
        sweep(12, 6, 51, 8192, 10)      # the horizon sweep's (12, 6) shape (run_random_linear.jl:80-108)
    if which in ("all", "quad"):
        quadruped(40, 2048, 8)
        quadruped(15, 2048, 8)
